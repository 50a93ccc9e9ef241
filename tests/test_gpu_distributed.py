"""
The N > 1 code paths on a one-GPU box.

RCCL itself with more than one rank needs more than one GPU (the driver's 8-GPU node).  What a one-GPU box can do:
  * a process group of ONE rank over `nccl`: every RCCL call the sharded paths make (init with a device id,
    all_gather, all_gather_into_tensor, all_reduce, barrier) executes on the card, through the same HIP runtime as
    libdlagp_hip.so;
  * two ranks on the one card over `gloo`: the by-sample partition with device tensors, and bench.py's exchanges.
Needs a real MI355X: `pytest -m gpu`.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from gpy_dla_detection_amd.samples import halton
from gpy_dla_detection_amd.synthetic import make_sightline
from tests.conftest import REPO, free_port, load_golden

pytestmark = pytest.mark.gpu


def run_bench(nproc, *args, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if nproc:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc,
               "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    else:
        cmd = [sys.executable]
    cmd += [os.path.join(REPO, "bench.py"), "--no-cpu-baseline", *args]
    res = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_collectives_run_over_rccl_with_one_rank():
    """
    bench.py launched as the driver launches it for N > 1 (torch.distributed.run, one process per GPU, `nccl`), with
    one rank: process-group set-up on the device, the all-gather of the evidences, the barrier and the max-reduce of
    the elapsed time all go through RCCL; every mode reproduces the evidence of a plain single-process run.
    """
    small = ["--steps", "3", "--warmup", "1", "--ramp-ms", "5", "--samples", "2000"]
    plain = run_bench(0, "--gpus", "1", *small)
    weak = run_bench(1, "--gpus", "1", *small)
    strong = run_bench(1, "--gpus", "1", "--strong", *small)
    for line in (weak, strong):
        assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["kernel_launches"] == 3
        assert line["evidence"] == plain["evidence"]
    assert weak["scaling"] == "weak" and strong["scaling"] == "strong"
    batch = ["--steps", "1", "--warmup", "1", "--ramp-ms", "5", "--samples", "1000", "--sightlines", "3"]
    for config in ("3", "4", "5"):
        alone = run_bench(0, "--gpus", "1", "--config", config, *batch)
        ranked = run_bench(1, "--gpus", "1", "--config", config, *batch)
        assert ranked["evidence"] == alone["evidence"] and ranked["value"] > 0


def test_two_ranks_on_one_gpu_rehearse_the_exchanges():
    """two processes on the one card, `gloo`: weak scaling doubles the evals, strong scaling keeps the evidence"""
    small = ["--steps", "3", "--warmup", "1", "--ramp-ms", "5", "--samples", "2000", "--rehearse-on-one-gpu"]
    plain = run_bench(0, "--gpus", "1", *small[:-1])
    weak = run_bench(2, "--gpus", "2", *small)
    strong = run_bench(2, "--gpus", "2", "--strong", *small)
    assert weak["n_gpus"] == 2 and weak["config"]["evals_per_step_per_gpu"] == 2000.0
    assert strong["config"]["evals_per_step_per_gpu"] == 1000.0
    # rank 0 sweeps the same sightline in all three runs; the split evaluation sums in another order
    assert weak["evidence"] == plain["evidence"]
    assert abs(strong["evidence"] - plain["evidence"]) <= 1e-11 * abs(plain["evidence"])


def test_plain_bench_invocation_launches_its_own_ranks():
    """
    `python bench.py --gpus 2` with NO launcher around it (how the driver has invoked bench.py so far): the parent
    starts torch.distributed.run itself as a child, before it touches the GPU, and relays rank 0's one line.
    """
    small = ["--steps", "3", "--warmup", "1", "--ramp-ms", "5", "--samples", "2000", "--rehearse-on-one-gpu"]
    plain = run_bench(0, "--gpus", "1", *small[:-1])
    own = run_bench(0, "--gpus", "2", *small)
    assert own["n_gpus"] == 2 and own["steps"] == 3 and own["value"] > 0
    assert own["config"]["evals_per_step_per_gpu"] == 2000.0 and own["evidence"] == plain["evidence"]
    strong = run_bench(0, "--gpus", "2", "--strong", *small)
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong"


def test_default_line_carries_the_other_configurations():
    """
    The default run (BASELINE configs[1]) also measures configs[2..4] at their one-GPU widths and the by-sample split,
    after the headline: each entry reproduces the evidence of its own `--config N` run, none reports an error.
    """
    line = run_bench(0, "--gpus", "1", "--steps", "20", "--warmup", "5")
    assert line["config"]["workload"].startswith("configs[1]") and line["roofline"]["kernel_launches"] == 20
    others = line["other_configs"]
    assert sorted(others) == ["3", "4", "5"]
    for key, entry in others.items():
        assert "error" not in entry, entry
        assert entry["ms_per_step"] > 0 and entry["evals_per_s"] > 1e6 and 0.2 < entry["accumulate_frac"] < 1.0
        alone = run_bench(0, "--gpus", "1", "--config", key, "--steps", "1", "--warmup", "1", "--ramp-ms", "5")
        assert entry["evidence"] == alone["evidence"]
    assert others["4"]["spectra_per_s"] > 0 and others["5"]["spectra_per_s"] > 0 and others["3"]["spectra_per_s"] is None
    strong = line["strong_n1"]
    assert "error" not in strong and strong["scaling"] == "strong" and strong["evidence"] == line["evidence"]
    # two ranks on the one card: the side measurements run their collectives on every rank
    two = run_bench(0, "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "5", "--warmup", "2", timeout=1100)
    assert two["n_gpus"] == 2 and sorted(two["other_configs"]) == ["3", "4", "5"]
    for entry in list(two["other_configs"].values()) + [two["strong_n2"]]:
        assert "error" not in entry and entry["n_gpus"] == 2
    assert two["other_configs"]["4"]["sightlines_per_gpu"] == 128


def test_bench_at_dr12q_sightline_lengths():
    """`--pixels N` (hidden): the batch configurations on synthetic sightlines of N unmasked pixels -- real DR12Q sightlines
    keep n <= 1250, median ~685 (reference set_parameters.py:33-34,58) -- with the line's own figures consistent with it"""
    line = run_bench(0, "--gpus", "1", "--config", "3", "--pixels", "685", "--sightlines", "8", "--steps", "2", "--warmup", "1",
                     "--ramp-ms", "5")
    assert "n=671 pixels (n_u=685)" in line["config"]["workload"] and line["config"]["sightlines_per_gpu"] == 8
    assert line["roofline"]["flop_per_eval"] == 671 * 20 * 21 + 2 * 671 * 20
    assert line["value"] > 1e6 and 0.2 < line["roofline"]["frac"] < 1.0 and "other_configs" not in line


def test_many_ranks_rehearsed_on_the_one_card():
    """
    The driver's multi-GPU run, rehearsed as far as a one-GPU box allows: `python bench.py --gpus 4 --rehearse-on-one-gpu
    --bind --steps 3`, invoked plainly -- the parent starts its own ranks as a child before anything touches the GPU -- with
    every rank on device 0 and gloo in place of RCCL.  ONE line, `n_gpus` 4, the headline and every side configuration
    measured on all ranks (each rank its own sightlines: weak scaling), the by-sample split over four ranks, the CPU
    binding reported.  Four ranks, not eight: the box's process guard allows at most six processes on the card at once
    (the eight-rank code path differs from this one in nothing but the world size; tests/test_host_logic.py runs the
    collectives' host logic with gloo at world sizes 2 and 4).
    """
    line = run_bench(0, "--gpus", "4", "--rehearse-on-one-gpu", "--bind", "--steps", "3", "--warmup", "1", timeout=1100)
    assert line["n_gpus"] == 4 and line["steps"] == 3 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["sightlines_per_gpu"] == 1 and line["config"]["evals_per_step_per_gpu"] == 10000.0
    assert sorted(line["other_configs"]) == ["3", "4", "5"]
    for entry in list(line["other_configs"].values()) + [line["strong_n4"]]:
        assert "error" not in entry, entry
        assert entry["n_gpus"] == 4 and entry["ms_per_step"] > 0
    assert line["other_configs"]["4"]["sightlines_per_gpu"] == 128 and line["other_configs"]["5"]["sightlines_per_gpu"] == 32
    assert line["strong_n4"]["scaling"] == "strong" and line["strong_n4"]["sightlines_per_gpu"] == 0.25
    assert "bound" in line["cpu_binding_rank0"]  # bound or not (a box without NUMA information): reported either way


WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["DLAGP_REPO"])
from gpy_dla_detection_amd import _lib, parallel
from gpy_dla_detection_amd.samples import halton
from gpy_dla_detection_amd.sightline import DeviceBatch, DeviceSightline
from gpy_dla_detection_amd.synthetic import make_sightline
from tests.conftest import load_golden

backend = os.environ["DLAGP_BACKEND"]
torch.cuda.set_device(0)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
out = os.environ["DLAGP_OUT"]

# by-spectrum shard + padded all-gather of the evidences (5 sightlines: 3 + 2 with two ranks)
sights = [make_sightline(i, n_unmasked=300 + 8 * i, n_masked=5) for i in range(5)]
table = parallel.model_evidences_sharded(sights, num_samples=64, max_dlas=2, device=0, rank=rank, world_size=world)
np.save(os.path.join(out, "evidences_rank%d.npy" % rank), table)

# by-sample split of ONE sightline, everything resident in HBM, one all-gather per level
g = load_golden("evidences_small.npz")
kwargs = {k[len("sight_"):]: int(g[k]) for k in g.files if k.startswith("sight_") and k != "sight_crc"}
sight = make_sightline(**kwargs)
S, max_dlas = int(g["S"]), g["dla_evidences"].shape[0]
nhi = 10.0 ** (20.0 + 3.0 * halton(S, 3))
batch = DeviceBatch([DeviceSightline.from_state(sight, device=0)])
uniforms = parallel.resampling_uniforms(1, max_dlas, S, seed=0)
ev, ll, base = parallel.sweep_sharded_samples_device(batch, g["dla_sample_z"][None, :], nhi, max_dlas,
                                                     float(g["dla_min_z_separation"]), uniforms,
                                                     rank=rank, world_size=world)
np.savez(os.path.join(out, "by_sample_rank%d.npz" % rank), ev=ev, ll=ll, base=base)
if world > 1:
    # a rank of a multi-rank job sweeping a sightline of its own: defaults (rank 0 of 1, no group) = local, no collective
    ev1, ll1, base1 = parallel.sweep_sharded_samples_device(batch, g["dla_sample_z"][None, :], nhi, max_dlas,
                                                            float(g["dla_min_z_separation"]), uniforms)
    np.savez(os.path.join(out, "local_rank%d.npz" % rank), ev=ev1, ll=ll1, base=base1)
# the catalogue driver sharded over the ranks: the per-quasar values of the whole catalogue on every rank after ONE all-gather
# (RCCL when the group is nccl), the per-sample arrays of a rank's own quasars written as its shard
from gpy_dla_detection_amd import run_bayes_select as rbs
from tests.test_host_logic import FixedPrior, catalogue_inputs
learned, spectra, z_qsos, params, dla_samples, sub_samples = catalogue_inputs(128)
spectra, z_qsos = list(spectra) * 3, [z + 0.001 * i for i, z in enumerate(list(z_qsos) * 3)]
res = rbs.process_qso(spectra, z_qsos, params, FixedPrior(), dla_samples, sub_samples, learned, max_dlas=3, batch_size=4,
                      device=0, rank=rank, world_size=world)
if world > 1:
    assert res["sample_log_likelihoods_dla"].shape[0] == len(res["local_rows"]) < len(spectra)
    assert res["log_posteriors_dla"].shape == (len(spectra), 3)
rbs.write_results(res, os.path.join(out, "catalogue.h5"))
with open(os.path.join(out, "runtime_rank%d.txt" % rank), "w") as f:
    f.write("\n".join(_lib.hip_runtime_copies()))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("backend,nproc", [("nccl", 1), ("gloo", 2)])
def test_sharded_sweeps_through_a_process_group(tmp_path, backend, nproc):
    """
    `parallel.model_evidences_sharded` (by spectrum) and `parallel.sweep_sharded_samples_device` (by sample, device
    tensors) inside a real process group: RCCL with one rank, and two ranks on the one card over gloo.  Both reproduce
    the single-process results; the by-sample split reproduces the reference's base indices exactly (golden G5).
    """
    from gpy_dla_detection_amd import parallel

    env = dict(os.environ, DLAGP_REPO=REPO, DLAGP_OUT=str(tmp_path), DLAGP_BACKEND=backend, MASTER_ADDR="127.0.0.1",
               OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)]
    res = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]

    sights = [make_sightline(i, n_unmasked=300 + 8 * i, n_masked=5) for i in range(5)]
    want = parallel.model_evidences_sharded(sights, num_samples=64, max_dlas=2, device=0)
    g = load_golden("evidences_small.npz")
    for rank in range(nproc):
        got = np.load(tmp_path / ("evidences_rank%d.npy" % rank))
        assert got.shape == (5, 4) and np.all(np.isfinite(got))
        np.testing.assert_allclose(got, want, rtol=1e-12)  # shards are batched differently: rounding only
        by_sample = np.load(tmp_path / ("by_sample_rank%d.npz" % rank))
        np.testing.assert_array_equal(by_sample["base"][0], g["dla_base_inds"])
        np.testing.assert_allclose(by_sample["ev"][0], g["dla_evidences"], rtol=1e-9)
        np.testing.assert_allclose(by_sample["ll"][0], g["dla_sample_ll"], rtol=1e-9, equal_nan=True)
        if nproc > 1:
            local = np.load(tmp_path / ("local_rank%d.npz" % rank))
            np.testing.assert_array_equal(local["base"][0], g["dla_base_inds"])
            np.testing.assert_allclose(local["ev"][0], g["dla_evidences"], rtol=1e-9)
        copies = (tmp_path / ("runtime_rank%d.txt" % rank)).read_text().split()
        assert len(copies) == 1, copies  # torch (RCCL) and the library share one HIP runtime
    # the sharded catalogue run: main file + shards merge into the single-process result (batched differently: rounding)
    from gpy_dla_detection_amd import run_bayes_select as rbs
    from tests.test_host_logic import FixedPrior, catalogue_inputs

    learned, spectra, z_qsos, params, dla_samples, sub_samples = catalogue_inputs(128)
    spectra, z_qsos = list(spectra) * 3, [z + 0.001 * i for i, z in enumerate(list(z_qsos) * 3)]
    alone = rbs.process_qso(spectra, z_qsos, params, FixedPrior(), dla_samples, sub_samples, learned, max_dlas=3, batch_size=4)
    main = str(tmp_path / "catalogue.npz")
    merged = (rbs.merge_results(main, [str(tmp_path / ("catalogue.rank%d.npz" % r)) for r in range(nproc)]) if nproc > 1
              else rbs._read_one(main))
    assert sorted(merged) == sorted(alone)
    np.testing.assert_array_equal(merged["base_sample_inds"], alone["base_sample_inds"])
    assert merged["base_sample_inds"].dtype == np.int32
    for name, value in alone.items():
        if np.asarray(value).dtype.kind == "f":
            np.testing.assert_allclose(merged[name], value, rtol=1e-10, equal_nan=True, err_msg=name)
