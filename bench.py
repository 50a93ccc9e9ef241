#!/usr/bin/env python3
"""
Benchmark of the GP model-evidence sweep on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C] [--strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads (SURVEY.md §8d "Configs -> inputs", numbered as there; BASELINE.json `configs` is the same list from 0).
Every sightline is synthetic: n = 3000 kept pixels (3061 unmasked, 3067 padded), rank k = 20, S = 10 000 Halton
(z_DLA, log N_HI) samples, 3 Lyman lines.  Samples, models and outputs are resident in HBM before the timed region.

  --config 2 (default; BASELINE configs[1], the configuration the metric is quoted on)
        per GPU 1 sightline x 10 000 DLA samples: one step = accumulate + finalize + log-mean-exp evidence
  --config 3 (BASELINE configs[2])   per GPU 64 sightlines: one step = DLA(1) sweep + sub-DLA sweep of the batch
  --config 4 (BASELINE configs[3] at its one-GPU width: 1024 spectra / 8 GPUs)
        per GPU 128 sightlines: one step = null evidences + sub-DLA sweep + DLA(1) sweep
  --config 5 (BASELINE configs[4] at its one-GPU width: 256 spectra / 8 GPUs)
        per GPU 32 sightlines: one step = the occupation sweep DLA(1..4), levels chained on the device

With N > 1 every rank sweeps its own sightlines (weak scaling, no data-path collective) and the per-step evidences
are all-gathered over RCCL once, at the end of the timed region.  `--strong` (config 2 only) instead splits the
10 000 samples of ONE sightline over the ranks: per step every rank evaluates its slice, one all-gather of S / N
log-likelihoods, and the evidence on every rank (the by-sample partition of SURVEY.md §8e).

`vs_baseline` is null: BASELINE.md holds no published number for this metric on this configuration (the reference's only
statement, "~4 mins for a single spectrum", is another size on unnamed hardware).

Prints ONE JSON line on rank 0: the driver's contract keys plus
  roofline      the accumulate kernels only: algorithmic FLOP of the timed launches / their summed HIP-event
                durations, against the dense FP64 MFMA peak; `frac_cold`: the same from 25 steps on an idle device
                (config 2), before the untimed clock ramp
  cpu_baseline  the oracle (NumPy/SciPy port of the reference loop) timed on this box's host cores on a bounded
                sample of the same workload (rank 0, N = 1 only): variant (iii) of BASELINE.md §3 as `value`,
                variants (i) and (ii) under `variants`
"""
import argparse
import json
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
MIN_Z_SEPARATION = 3000.0e3 / 299792458.0

# sightlines per GPU, and what one step sweeps
WORKLOADS = {
    2: dict(sightlines=1, baseline_index=1, what="10 000 DLA samples of 1 sightline: accumulate + finalize + evidence"),
    3: dict(sightlines=64, baseline_index=2, what="DLA(1) sweep + sub-DLA sweep of 64 sightlines"),
    4: dict(sightlines=128, baseline_index=3, what="null + sub-DLA + DLA(1) of 128 sightlines (1024 spectra / 8 GPUs)"),
    5: dict(sightlines=32, baseline_index=4, what="occupation sweep DLA(1..4) of 32 sightlines (256 spectra / 8 GPUs)"),
}


def algorithmic_flops_per_eval(n: int, k: int) -> float:
    """symmetric Gram n k (k+1) + projection 2 n k"""
    return float(n * k * (k + 1) + 2 * n * k)


def measured_hbm_traffic(config: int):
    """
    HBM bytes per accumulate-kernel launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, collected in
    their own rocprofv3 runs of this same command, gfx950 read-side correction applied).  None when absent.
    """
    for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                data = json.load(f)
            if int(data.get("config", 2)) == config:
                return float(data["hbm_bytes_per_launch"]), name
        except (OSError, KeyError, ValueError):
            pass
    return None, None


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle on the host cores (test infrastructure used as the reported baseline, never shipped)
# ------------------------------------------------------------------------------------------------
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
        ref_like.sample_log_likelihood_k_dlas(st, np.atleast_1d(zi), np.atleast_1d(ni))
    return len(zs)


def _single_process_rate(evals, seconds):
    """evals/s of this process looping over `evals` = [(z tuple, nhi tuple), ...] for about `seconds`"""
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        _cpu_worker_eval(([evals[done % len(evals)][0]], [evals[done % len(evals)][1]]))
        done += 1
    return done / (time.perf_counter() - t0), done


def cpu_baseline(sight, evals, cpu_seconds: float = 18.0, single_seconds: float = 4.0):
    """
    BASELINE.md §3 on a bounded sample of the step's evals (`evals`: parameter sets drawn from the workload):
      (iii) one process per core of this job's share (16 on a one-GPU box), BLAS pinned to 1 thread: `value`
      (i)   1 process x 1 BLAS thread, (ii) 1 process x default BLAS threads: under `variants`
    About cpu_seconds + 2 x single_seconds of CPU work in total.
    """
    import multiprocessing as mp

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(cores, 16))  # a one-GPU box gives this job a 16-core share
    ctx = mp.get_context("fork")
    with ctx.Pool(workers, initializer=_cpu_worker_init, initargs=(sight,)) as pool:
        pool.map(_cpu_worker_eval, [([evals[0][0]], [evals[0][1]])] * workers)  # imports, BLAS init
        done, elapsed, cursor = 0, 0.0, 0
        while elapsed * workers < cpu_seconds:
            chunks = []
            for w in range(workers):
                take = [evals[(cursor + w * 128 + i) % len(evals)] for i in range(128)]
                chunks.append(([t[0] for t in take], [t[1] for t in take]))
            cursor += 128 * workers
            t0 = time.perf_counter()
            done += sum(pool.map(_cpu_worker_eval, chunks))
            elapsed += time.perf_counter() - t0
    variants = {}
    _cpu_worker_init(sight)  # this process: BLAS limited to one thread ...
    rate_i, n_i = _single_process_rate(evals, single_seconds)
    variants["i_one_process_one_blas_thread"] = {"value": rate_i, "cores": 1, "evals": n_i}
    try:  # ... and with the default thread count
        from threadpoolctl import threadpool_info, threadpool_limits

        default_threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
        with threadpool_limits(limits=cores):
            rate_ii, n_ii = _single_process_rate(evals, single_seconds)
        variants["ii_one_process_default_blas_threads"] = {"value": rate_ii, "cores": cores, "evals": n_ii,
                                                           "blas_threads_seen": default_threads}
    except Exception:
        pass
    return {
        "value": done / elapsed,
        "unit": "evals/s",
        "cores": workers,
        "kind": "port",
        "sample": "%d evals drawn from the %d parameter sets of one step's workload, %d processes x 1 BLAS thread, %.1f s wall"
                  % (done, len(evals), workers, elapsed),
        "variants": variants,
        "host_cpus": os.cpu_count(),
    }


def baseline_sample(config, sight, z, nhi_dla, nhi_sub):
    """parameter sets representative of one step of `config` (one sightline's share)"""
    S = len(z)
    pick = np.arange(0, S, max(S // 512, 1))[:512]
    if config == 2:
        return [((z[i],), (nhi_dla[i],)) for i in pick]
    if config in (3, 4):
        return [((z[i],), (nhi_dla[i],)) for i in pick[::2]] + [((z[i],), (nhi_sub[i],)) for i in pick[1::2]]
    evals = []  # config 5: levels 1..4 in equal numbers; deeper absorbers drawn from the same QMC points
    for j, i in enumerate(pick):
        kd = 1 + j % 4
        members = [(i + 97 * m) % S for m in range(kd)]
        evals.append((tuple(z[members]), tuple(nhi_dla[members])))
    return evals


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--strong", action="store_true", help="config 2: split ONE sightline's samples over the ranks")
    ap.add_argument("--ramp-ms", type=float, default=150.0, help=argparse.SUPPRESS)  # untimed clock ramp before the warmup
    ap.add_argument("--samples", type=int, default=NUM_SAMPLES, help=argparse.SUPPRESS)
    ap.add_argument("--sightlines", type=int, default=0, help=argparse.SUPPRESS)  # tuning: override the batch width
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--splits", type=int, default=0, help=argparse.SUPPRESS)  # tuning: force the pixel-split count
    ap.add_argument("--option", action="append", default=[], help=argparse.SUPPRESS)  # tuning: library option name=value
    ap.add_argument("--debug", type=int, default=0, help=argparse.SUPPRESS)  # 2: per-phase cycle stamps (diagnostic)
    # rehearsal of the N > 1 code path on a one-GPU box: every rank on device 0, gloo instead of RCCL
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    config = args.config
    if args.strong and config != 2:
        raise SystemExit("--strong splits the samples of configs[1]'s single sightline: use --config 2")
    steps = args.steps if args.steps is not None else (200 if config == 2 else 5)
    warmup = args.warmup if args.warmup is not None else (20 if config == 2 else 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    # launched through torch.distributed.run: the process group and every collective below run even with one rank
    # (a one-GPU box then exercises the RCCL calls of the N > 1 path); plain `python bench.py` has no group
    dist_on = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)

    from gpy_dla_detection_amd import sightline as sl
    from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline

    S = args.samples
    nb = args.sightlines or WORKLOADS[config]["sightlines"]
    first = 0 if args.strong else rank * nb  # weak scaling: every rank has its own sightlines
    sights = [make_sightline(first + i, n_unmasked=N_UNMASKED, n_masked=N_MASKED, k=RANK_K) for i in range(nb)]
    z_host = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi_dla_host = make_qmc_samples(sights[0], S)[2]
    nhi_sub_host = make_qmc_samples(sights[0], S, sub_dla=True)[2]

    # the CPU baseline forks worker processes: do it before this process touches the GPU
    baseline = None
    if world == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        baseline = cpu_baseline(sights[0], baseline_sample(config, sights[0], z_host[0], nhi_dla_host, nhi_sub_host))

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
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    devs = [sl.DeviceSightline.from_state(s, device=local_rank) for s in sights]
    batch = sl.DeviceBatch(devs) if config != 2 else None
    if args.splits:
        sl.set_option("pixel_splits", args.splits)
    if args.debug:
        sl.set_option("debug", args.debug)
    for item in args.option:
        name, value = item.split("=")
        sl.set_option(name, int(value))

    f64 = dict(dtype=torch.float64, device=device)
    z = torch.from_numpy(z_host).to(device)
    nhi_dla = torch.from_numpy(nhi_dla_host).to(device)
    nhi_sub = torch.from_numpy(nhi_sub_host).to(device)
    max_dlas = 4 if config == 5 else 1
    # Every step's evidences land in their own slot; the ranks exchange them with ONE all-gather at the end of the
    # timed region -- the communication pattern of a sharded catalogue run (parallel.model_evidences_sharded).
    per_step = {2: 1, 3: 2 * nb, 4: 3 * nb, 5: 4 * nb}[config]
    slots = max(steps, warmup, 1)
    evidences = torch.zeros(slots * per_step, **f64)
    gathered = torch.zeros(world * slots * per_step, **f64) if dist_on else None
    counter = [0]

    if config == 2:
        lo, hi = (0, S)
        if args.strong:
            per_rank = -(-S // world)
            lo, hi = min(rank * per_rank, S), min((rank + 1) * per_rank, S)
            local = torch.zeros(per_rank, **f64)
            everything = torch.zeros(world * per_rank, **f64)
        out = torch.empty(S, **f64)
        nhi_col = nhi_dla.contiguous()

        def step():
            slot = counter[0] % slots
            counter[0] += 1
            stream = torch.cuda.current_stream().cuda_stream
            if not args.strong:
                devs[0].eval_device(S, 1, z.data_ptr(), nhi_col.data_ptr(), out.data_ptr(), stream)
                target = out
            else:  # this rank's slice, then all S log-likelihoods on every rank (S / N doubles per rank over RCCL)
                devs[0].eval_device(hi - lo, 1, z.data_ptr() + 8 * lo, nhi_col.data_ptr() + 8 * lo, local.data_ptr(), stream)
                if dist_on and args.rehearse_on_one_gpu:
                    dist.all_gather(list(everything.view(world, -1).unbind(0)), local)
                elif dist_on:
                    dist.all_gather_into_tensor(everything, local)
                else:
                    everything.copy_(local)
                target = everything
            # "- log S" and the log-mean-exp model evidence of the level (reference dla_gp.py:155-190)
            sl.level_evidence_device(target.data_ptr(), 1, S, 0, evidences.data_ptr() + 8 * slot, device=local_rank,
                                     stream=stream)
    else:
        from gpy_dla_detection_amd import parallel

        uniforms = torch.from_numpy(parallel.resampling_uniforms(nb, max_dlas, S, seed=0)).to(device) if max_dlas > 1 else None
        sample_ll = torch.empty((nb, max_dlas, S), **f64)
        base = torch.zeros((nb, max(max_dlas - 1, 1), S), dtype=torch.int32, device=device)

        def step():
            slot = counter[0] % slots
            counter[0] += 1
            stream = torch.cuda.current_stream().cuda_stream
            ev = evidences.data_ptr() + 8 * slot * per_step
            if config == 5:
                batch.sweep_device(S, 4, z.data_ptr(), nhi_dla.data_ptr(), 0, MIN_Z_SEPARATION, uniforms.data_ptr(), ev,
                                   sample_ll.data_ptr(), base.data_ptr(), stream)
                return
            batch.sweep_device(S, 1, z.data_ptr(), nhi_sub.data_ptr(), 0, MIN_Z_SEPARATION, 0, ev, sample_ll.data_ptr(), 0, stream)
            batch.sweep_device(S, 1, z.data_ptr(), nhi_dla.data_ptr(), 0, MIN_Z_SEPARATION, 0, ev + 8 * nb,
                               sample_ll.data_ptr(), 0, stream)
            if config == 4:  # NullGP.log_model_evidence of every member: one launch, host-pointer entry
                evidences[slot * per_step + 2 * nb: slot * per_step + 3 * nb] = torch.from_numpy(batch.nulls()).to(device)

    def exchange():
        if dist_on and args.rehearse_on_one_gpu:  # gloo has no all_gather_into_tensor for device tensors
            dist.all_gather(list(gathered.view(world, -1).unbind(0)), evidences)
        elif dist_on:
            dist.all_gather_into_tensor(gathered, evidences)

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(count):
        sl.set_option("timing", 1)
        sl.timing_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(count):
            step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        sl.set_option("timing", 0)
        kernel_ms, launches = sl.timing_read(reset=True)
        return elapsed, kernel_ms, launches

    flops = algorithmic_flops_per_eval(sights[0].n, sights[0].k)
    evals_per_step = {2: float(hi - lo) if config == 2 and args.strong else float(S), 3: 2.0 * nb * S,
                      4: nb * (1.0 + 2 * S), 5: 4.0 * nb * S}[config]
    mfma_evals_per_step = {2: evals_per_step, 3: 2.0 * nb * S, 4: 2.0 * nb * S, 5: 4.0 * nb * S}[config]

    # Cold figure: the device leaves its idle power state only after tens of milliseconds of load, so the first ~40
    # launches after idle run ~12 % slower than the sustained rate.  One untimed step creates the launch plan and
    # the scratch buffers, the device idles again, and 25 steps are timed with no ramp in front of them.
    frac_cold = None
    if config == 2 and not args.strong:
        step()
        torch.cuda.synchronize()
        time.sleep(0.5)
        _, cold_ms, cold_launches = timed(25)
        if cold_launches:
            frac_cold = S * flops / (cold_ms / 1e3 / cold_launches) / 1e12 / PEAK_FP64_MFMA_TFLOPS

    # Untimed steps until the device has been busy for --ramp-ms, then the W warmup steps the contract asks for, then
    # exactly K timed steps.  (local work only when not --strong -- the loop is time-based, so ranks may run different
    # numbers of iterations; --strong communicates per step, so there the ramp is a fixed number of steps)
    if args.strong and dist_on:
        for _ in range(60):
            step()
        torch.cuda.synchronize()
    else:
        ramp_start = time.perf_counter()
        while (time.perf_counter() - ramp_start) * 1e3 < args.ramp_ms:
            for _ in range(10 if config == 2 else 1):
                step()
            torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    exchange()
    fence()
    counter[0] = 0
    sl.set_option("timing", 1)
    sl.timing_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    exchange()
    fence()
    elapsed = time.perf_counter() - t0
    sl.set_option("timing", 0)
    kernel_ms, launches = sl.timing_read(reset=True)
    if args.debug & 2:
        st = sl.debug_read()
        if rank == 0 and st[5]:
            names = ["voigt", "conv", "mfma_loop", "epilogue", "passes+epilogue", None, "prologue"]
            print("stamps: waves=%d  " % st[5] + "  ".join("%s=%.0f" % (n, st[i] / st[5]) for i, n in enumerate(names) if n) + "  (cycles per wave)", file=sys.stderr)

    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank holds every rank's evidences after the exchange: rank r's slots must equal its own
        mine = gathered.view(world, -1)[rank]
        if not torch.equal(torch.nan_to_num(mine), torch.nan_to_num(evidences)):
            raise SystemExit("rank %d: the gathered evidences differ from the local ones" % rank)

    if rank == 0:
        total_evals = (float(S) if args.strong else evals_per_step * world) * steps
        value = total_evals / elapsed
        kernel_s_total = kernel_ms / 1e3
        # timed accumulate launches of this rank: all MFMA-path launches of the step (config 4's null launch is the
        # generic kernel on one sample per sightline and is timed with them; its FLOPs are not counted)
        achieved = mfma_evals_per_step * steps * flops / kernel_s_total / 1e12 if kernel_s_total > 0 else 0.0
        traffic, traffic_file = measured_hbm_traffic(config)
        line = {
            "metric": "QMC log-likelihood evals/sec",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "configs[%d] (SURVEY 8d config %d): %s; S=%d QMC samples, n=%d pixels (n_u=%d), rank k=%d, 3 Lyman lines"
                            % (WORKLOADS[config]["baseline_index"], config, WORKLOADS[config]["what"], S, sights[0].n,
                               N_UNMASKED, sights[0].k),
                "sightlines_per_gpu": nb if not args.strong else 1.0 / world,
                "samples_per_sightline": S,
                "evals_per_step_per_gpu": evals_per_step,
                "parallelism": ("samples of one sightline split across GPUs, one all-gather of S/N log-likelihoods per step"
                                if args.strong else
                                "sightlines sharded across GPUs, one all-gather of the evidences at the end"),
            },
            "spectra_per_s": nb * world * steps / elapsed if config in (4, 5) else None,
            "spectra_per_s_equiv": value / (1 + S + 4 * S),
            "roofline": {
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_FP64_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                "frac_cold": frac_cold,
                "traffic": traffic if S == NUM_SAMPLES else None,
                "traffic_unit": "HBM bytes per accumulate launch (PMC, profiles/%s)" % traffic_file if traffic_file else None,
                "kernel": "accumulate_mfma (all instantiations launched in the timed region)",
                "kernel_ms": kernel_ms / max(launches, 1),
                "kernel_launches": launches,
                "flop_per_eval": flops,
            },
            "evidence": float(evidences[0].item()),
        }
        if baseline is not None:
            line["cpu_baseline"] = baseline
        print(json.dumps(line), flush=True)

    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
