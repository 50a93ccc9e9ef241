#!/usr/bin/env python3
"""
Benchmark of the GP model-evidence sweep on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): per GPU one synthetic sightline of
n = 3000 kept pixels (3061 unmasked, 3067 padded), rank k = 20, and S = 10 000 Halton
(z_DLA, log N_HI) samples with one absorber each.  One step = one pass of the hot path over
that batch: 10 000 x [Voigt-absorbed model -> low-rank Gaussian log-density] followed by the
log-mean-exp model evidence on the device.  Samples, model and outputs are resident in HBM
before the timed region.  With N > 1 every rank sweeps its own sightline (weak scaling,
no data-path collective) and the per-step evidences are all-gathered over RCCL once, at the end of the
timed region.

Prints ONE JSON line on rank 0: the driver's contract keys plus
  roofline      accumulate kernel only: algorithmic FLOP/launch / mean HIP-event duration of
                that kernel inside the timed region, against the dense FP64 MFMA peak
  cpu_baseline  the oracle (NumPy/SciPy port of the reference loop) timed on this box's host
                cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X datasheet, dense FP64 matrix (= FP64 vector) rate
NUM_SAMPLES = 10000
N_UNMASKED, N_MASKED, RANK_K = 3061, 61, 20


def algorithmic_flops_per_eval(n: int, k: int) -> float:
    """symmetric Gram n k (k+1) + projection 2 n k"""
    return float(n * k * (k + 1) + 2 * n * k)


def measured_hbm_traffic():
    """
    HBM bytes per accumulate-kernel launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE,
    collected in their own rocprofv3 runs of this same command, gfx950 read-side correction applied):
    profiles/r01_pmc_traffic.json.  None when the file is absent.
    """
    path = os.path.join(REPO, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(sight, z, nhi, cpu_seconds: float = 24.0):
    """
    The oracle (NumPy/SciPy port of the reference's per-sample loop) timed on the host cores: one
    process per core of this job's share (16 on a one-GPU box), BLAS pinned to 1 thread, on a bounded
    sample of the step's evals (about `cpu_seconds` of CPU time in total; the sample wraps around
    the 10 000 evals of the step if the cores are fast).
    """
    import multiprocessing as mp

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(cores, 16))  # a one-GPU box gives this job a 16-core share
    ctx = mp.get_context("fork")
    with ctx.Pool(workers, initializer=_cpu_worker_init, initargs=(sight,)) as pool:
        # warm the workers (imports, BLAS init), then time rounds of 32 evals per worker until the
        # CPU-time budget is spent
        pool.map(_cpu_worker_eval, [(z[:2], nhi[:2])] * workers)
        done, elapsed, cursor = 0, 0.0, 0
        while elapsed * workers < cpu_seconds:
            idx = [np.arange(cursor + w * 32, cursor + (w + 1) * 32) % len(z) for w in range(workers)]
            cursor += 32 * workers
            t0 = time.perf_counter()
            done += sum(pool.map(_cpu_worker_eval, [(z[i], nhi[i]) for i in idx]))
            elapsed += time.perf_counter() - t0
    return {
        "value": done / elapsed,
        "unit": "evals/s",
        "cores": workers,
        "kind": "port",
        "sample": "%d evals drawn from the %d of one step, %d processes x 1 BLAS thread, %.1f s wall" % (done, len(z), workers, elapsed),
    }


_CPU_STATE = None


def _cpu_worker_init(sight):
    global _CPU_STATE
    try:
        from threadpoolctl import threadpool_limits

        threadpool_limits(1)
    except Exception:
        pass
    from oracle import ref_like

    _CPU_STATE = (
        ref_like,
        ref_like.SpectrumState(sight.y, sight.v, sight.this_mu, sight.this_M, sight.this_omega2,
                               sight.padded_wavelengths, sight.unmasked_wavelengths, sight.keep),
    )


def _cpu_worker_eval(args):
    zs, ns = args
    ref_like, st = _CPU_STATE
    for zi, ni in zip(zs, ns):
        ref_like.sample_log_likelihood_k_dlas(st, [zi], [ni])
    return len(zs)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-ms", type=float, default=150.0, help=argparse.SUPPRESS)  # untimed clock ramp before the warmup
    ap.add_argument("--samples", type=int, default=NUM_SAMPLES, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--splits", type=int, default=0, help=argparse.SUPPRESS)  # tuning: force the pixel-split count
    ap.add_argument("--option", action="append", default=[], help=argparse.SUPPRESS)  # tuning: library option name=value
    ap.add_argument("--debug", type=int, default=0, help=argparse.SUPPRESS)  # 2: per-phase cycle stamps (diagnostic)
    # rehearsal of the N > 1 code path on a one-GPU box: every rank on device 0, gloo instead of RCCL
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))

    from gpy_dla_detection_amd import sightline as sl
    from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline

    S = args.samples
    sight = make_sightline(rank, n_unmasked=N_UNMASKED, n_masked=N_MASKED, k=RANK_K)
    z_host, _, nhi_host = make_qmc_samples(sight, S)

    # the CPU baseline forks worker processes: do it before this process touches the GPU
    baseline = None
    if world == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        baseline = cpu_baseline(sight, z_host, nhi_host)

    import torch
    import torch.distributed as dist

    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch --gpus %d through torch.distributed.run (one process per GPU)" % args.gpus)
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    dev = sl.DeviceSightline.from_state(sight, device=local_rank)
    if args.splits:
        sl.set_option("pixel_splits", args.splits)
    if args.debug:
        sl.set_option("debug", args.debug)
    for item in args.option:
        name, value = item.split("=")
        sl.set_option(name, int(value))
    z = torch.from_numpy(z_host).to(device)
    nhi = torch.from_numpy(nhi_host).to(device)
    out = torch.empty(S, dtype=torch.float64, device=device)
    # Every step's evidence lands in its own slot; the ranks exchange them with ONE all-gather at the end of
    # the timed region -- the communication pattern of a sharded catalogue run (parallel.model_evidences_sharded:
    # each rank sweeps its sightlines, one gather of the per-sightline evidences at the end).
    slots = max(args.steps, args.warmup, 1)
    evidences = torch.zeros(slots, dtype=torch.float64, device=device)
    gathered = torch.zeros(world * slots, dtype=torch.float64, device=device) if world > 1 else None
    counter = [0]

    def step():
        slot = counter[0] % slots
        counter[0] += 1
        stream = torch.cuda.current_stream().cuda_stream
        dev.eval_device(S, 1, z.data_ptr(), nhi.data_ptr(), out.data_ptr(), stream)
        # "- log S" and the log-mean-exp model evidence of the level (reference dla_gp.py:155-190)
        sl.level_evidence_device(out.data_ptr(), 1, S, 0, evidences.data_ptr() + 8 * slot, device=local_rank,
                                 stream=stream)

    def exchange():
        if world > 1 and args.rehearse_on_one_gpu:  # gloo has no all_gather_into_tensor for device tensors
            dist.all_gather(list(gathered.view(world, slots).unbind(0)), evidences)
        elif world > 1:
            dist.all_gather_into_tensor(gathered, evidences)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The GPU leaves its idle power state only after tens of milliseconds of load: the first ~40 steps run
    # ~12 % slower than the sustained rate (profiles/r01_clock_ramp.txt).  Untimed steps until the device
    # has been busy for --ramp-ms, then the W warmup steps the contract asks for, then exactly K timed steps.
    # (local work only -- step() never communicates: the loop is time-based, so ranks may run different numbers of
    # iterations)
    ramp_start = time.perf_counter()
    while (time.perf_counter() - ramp_start) * 1e3 < args.ramp_ms:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    exchange()
    fence()
    counter[0] = 0
    sl.set_option("timing", 1)
    sl.timing_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    exchange()
    fence()
    elapsed = time.perf_counter() - t0
    sl.set_option("timing", 0)
    kernel_ms, launches = sl.timing_read(reset=True)
    if args.debug & 2:
        st = sl.debug_read()
        if rank == 0 and st[5]:
            names = ["voigt", "conv", "mfma_loop", "epilogue", "wave_total"]
            print("stamps: waves=%d  " % st[5] + "  ".join("%s=%.0f" % (n, st[i] / st[5]) for i, n in enumerate(names)) + "  (cycles per wave)", file=sys.stderr)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        evals = float(S) * args.steps * world
        value = evals / elapsed
        flops = algorithmic_flops_per_eval(sight.n, sight.k)
        kernel_s = kernel_ms / 1e3 / max(launches, 1)
        achieved = S * flops / kernel_s / 1e12 if kernel_s > 0 else 0.0
        line = {
            "metric": "QMC log-likelihood evals/sec",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: 1 sightline x %d QMC DLA samples per GPU, n=%d pixels (n_u=%d), rank k=%d, 3 Lyman lines, 1 absorber" % (S, sight.n, N_UNMASKED, sight.k),
                "sightlines_per_gpu": 1,
                "samples_per_sightline": S,
                "parallelism": "sightlines sharded across GPUs, one all-gather of the evidences at the end",
            },
            "spectra_per_s_equiv": value / (1 + S + 4 * S),
            "roofline": {
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_FP64_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                "traffic": measured_hbm_traffic() if S == NUM_SAMPLES else None,
                "traffic_unit": "bytes per launch (PMC, profiles/r01_pmc_traffic.json)",
                "kernel": "accumulate",
                "kernel_ms": kernel_s * 1e3,
                "flop_per_eval": flops,
                "logical_hbm_frac": (S * 8.0 * sight.n * (sight.k + 5) / kernel_s / 8e12) if kernel_s > 0 else 0.0,
            },
            "evidence": float(evidences[0].item()),
        }
        if baseline is not None:
            line["cpu_baseline"] = baseline
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
