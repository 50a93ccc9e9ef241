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


def measured_hbm_traffic(config: int, n: int, S: int, profiles_dir: str = None):
    """
    HBM bytes per accumulate-kernel launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, collected in
    their own rocprofv3 runs of this same command, gfx950 read-side correction applied) -- only from a file that was
    collected on THIS workload: same configuration, same kept pixels n, same sample count S (a file that does not say
    counts for nothing).  (None, None) otherwise: a run on other pixels or samples has no measured traffic to report.
    """
    profiles_dir = profiles_dir or os.path.join(REPO, "profiles")
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json",
                 "r02_pmc_traffic.json"):
        try:
            with open(os.path.join(profiles_dir, name)) as f:
                data = json.load(f)
            if (int(data["config"]), int(data["n"]), int(data["S"])) == (int(config), int(n), int(S)):
                return float(data["hbm_bytes_per_launch"]), name
        except (OSError, KeyError, ValueError, TypeError):
            pass
    return None, None


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle on the host cores (test infrastructure used as the reported baseline, never shipped)
# ------------------------------------------------------------------------------------------------
_CPU_STATE = None


def _cpu_worker_init(sight, core_queue=None):
    global _CPU_STATE
    if core_queue is not None:  # one worker per core: pinned, so that chunks are not migrated or stacked
        try:
            os.sched_setaffinity(0, {core_queue.get()})
        except (OSError, AttributeError):
            pass
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
    t0 = time.perf_counter()
    for zi, ni in zip(zs, ns):
        ref_like.sample_log_likelihood_k_dlas(st, np.atleast_1d(zi), np.atleast_1d(ni))
    return len(zs), time.perf_counter() - t0


def _single_process_rate(evals, seconds):
    """evals/s of this process looping over `evals` = [(z tuple, nhi tuple), ...] for about `seconds`"""
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        _cpu_worker_eval(([evals[done % len(evals)][0]], [evals[done % len(evals)][1]]))
        done += 1
    return done / (time.perf_counter() - t0), done


def host_cpu_share():
    """
    (cores to use, how that number was arrived at).  The affinity mask of a one-GPU box of the pool shows every CPU of
    the 8-GPU host; what this job may actually use is its cgroup's CPU quota when one is set, and otherwise the
    pool's stated share of 16 cores per GPU (worker pools beyond it are killed by the box's process guard).  One
    worker per PHYSICAL core: SMT siblings of a core already taken are skipped.
    """
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    physical, seen = [], set()
    for cpu in allowed:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu) as f:
                key = f.read().strip()
        except OSError:
            key = str(cpu)
        if key not in seen:
            seen.add(key)
            physical.append(cpu)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            limit, period = f.read().split()[:2]
        if limit != "max":
            quota = float(limit) / float(period)
    except (OSError, ValueError):
        pass
    physical = _idlest_first(physical)
    if quota is not None:
        count = max(1, min(len(physical), int(quota)))
        why = "cgroup cpu.max quota %.1f CPUs (affinity mask: %d CPUs, %d physical cores)" % (quota, len(allowed), len(physical))
    elif len(physical) > 16:
        count = 16
        why = ("the pool's share of 16 cores per GPU (affinity mask: %d CPUs = the whole 8-GPU host, %d physical cores, "
               "no cgroup quota)" % (len(allowed), len(physical)))
    else:
        count = len(physical)
        why = "every physical core of the affinity mask (%d CPUs)" % len(allowed)
    return physical[:count], why


def _idlest_first(cpus, interval: float = 0.25):
    """
    `cpus` ordered by how busy each was over `interval` seconds (/proc/stat), idlest first, ties in index order: the
    host is shared with the jobs of its other GPUs, and pinning to cores another tenant is using is what makes a
    baseline irreproducible.
    """
    def snapshot():
        busy = {}
        try:
            with open("/proc/stat") as f:
                for ln in f:
                    if ln.startswith("cpu") and ln[3].isdigit():
                        parts = ln.split()
                        ticks = [int(x) for x in parts[1:9]]
                        busy[int(parts[0][3:])] = sum(ticks) - ticks[3] - ticks[4]  # all but idle and iowait
        except (OSError, ValueError, IndexError):
            pass
        return busy

    before = snapshot()
    time.sleep(interval)
    after = snapshot()
    return sorted(cpus, key=lambda c: (after.get(c, 0) - before.get(c, 0), c))


def host_description():
    info = {"numpy": np.__version__}
    try:
        import scipy

        info["scipy"] = scipy.__version__
    except ImportError:
        pass
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    info["cpu_model"] = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        from threadpoolctl import threadpool_info

        import scipy.linalg  # noqa: F401  (maps scipy's BLAS so that it is listed)

        info["blas"] = sorted({"%s %s (%s, %s)" % (p.get("internal_api"), p.get("version"), p.get("threading_layer"),
                                                  p.get("architecture")) for p in threadpool_info()
                               if p.get("user_api") == "blas"})
    except Exception:
        pass
    return info


def _default_threads_child(conn, sight, evals, seconds, cores):
    """
    Variant (ii) in a FRESH interpreter (multiprocessing "spawn"): this process has never limited its BLAS pools, so
    "default threads" is what a user who simply runs the reference gets on this job's share of the host -- the
    affinity mask is narrowed to that share BEFORE NumPy is imported, which is where OpenBLAS sizes its pool.
    """
    try:
        try:
            os.sched_setaffinity(0, set(cores))
        except (OSError, AttributeError):
            pass
        global _CPU_STATE
        from oracle import ref_like

        _CPU_STATE = (ref_like, ref_like.SpectrumState(sight.y, sight.v, sight.this_mu, sight.this_M, sight.this_omega2,
                                                       sight.padded_wavelengths, sight.unmasked_wavelengths, sight.keep))
        _single_process_rate(evals, 0.5)  # imports, BLAS start-up
        rate, done = _single_process_rate(evals, seconds)
        threads = 1
        try:
            from threadpoolctl import threadpool_info

            threads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] + [1])
        except Exception:
            pass
        conn.send({"value": rate, "evals": done, "blas_threads_used": threads})
    except Exception as exc:  # noqa: BLE001 -- reported, never fatal for the bench line
        conn.send({"error": "%s: %s" % (type(exc).__name__, exc)})
    finally:
        conn.close()


def cpu_baseline(sight, evals, chunks: int = 5, chunk_seconds: float = 2.0, single_seconds: float = 3.0):
    """
    BASELINE.md §3 on a bounded sample of the step's evals (`evals`: parameter sets drawn from the workload):
      (iii) one process per core of this job's share (`host_cpu_share`), each PINNED to its own physical core, BLAS
            limited to 1 thread: `value` = the MEDIAN over `chunks` equal timed chunks of (evals of the chunk / wall
            time of the chunk), after ONE untimed chunk of the same size (the first chunk of a run is a ramp: page
            faults of the workers' first full pass, the cores' clocks -- round 4's driver line read 17 / 22 / 27 / 26 / 25 k
            where the committed one read 27-30 k throughout); per-chunk rates, the untimed chunk's rate, the per-core
            rate from the workers' own clocks and the spread are listed
      (i)   1 process x 1 BLAS thread (this process, limited for the duration of the measurement only),
      (ii)  1 process x default BLAS threads, measured in a FRESH interpreter whose affinity is this job's share: the
            thread count reported is the one its BLAS actually ran with
    About (chunks + 2 ... 4) x chunk_seconds + 2 x single_seconds + start-up of wall time in total.
    """
    import multiprocessing as mp

    cores, why = host_cpu_share()
    workers = len(cores)
    ctx = mp.get_context("fork")
    queue = ctx.Queue()
    for core in cores:
        queue.put(core)

    def chunk_args(per_worker, cursor):
        out = []
        for w in range(workers):
            take = [evals[(cursor + w * per_worker + i) % len(evals)] for i in range(per_worker)]
            out.append(([t[0] for t in take], [t[1] for t in take]))
        return out

    with ctx.Pool(workers, initializer=_cpu_worker_init, initargs=(sight, queue)) as pool:
        pool.map(_cpu_worker_eval, chunk_args(1, 0), chunksize=1)  # imports, BLAS init
        # size a chunk from a short calibration: the same number of evals for every worker and every chunk
        t0 = time.perf_counter()
        pool.map(_cpu_worker_eval, chunk_args(16, 0), chunksize=1)
        per_worker = max(16, int(16 * chunk_seconds / max(time.perf_counter() - t0, 1e-3)))
        rates, busy, done, cursor, untimed_rates = [], [], 0, 0, []

        def one_chunk():
            nonlocal cursor
            t0 = time.perf_counter()
            results = pool.map(_cpu_worker_eval, chunk_args(per_worker, cursor), chunksize=1)
            wall = time.perf_counter() - t0
            cursor += per_worker * workers
            return results, sum(r[0] for r in results) / wall

        # the ramp: untimed chunks until two in a row agree within 5 % (at most four) -- on the pool's hosts the first two
        # or three chunks of a run climb (page faults of the workers' first passes, the cores' clocks, other tenants' load
        # moving off the pinned cores): 21.0 / 21.9 / 26.4 / 26.3 / 28.3 k in one collection of this round
        for _ in range(4):
            untimed_rates.append(one_chunk()[1])
            if len(untimed_rates) >= 2 and abs(untimed_rates[-1] - untimed_rates[-2]) <= 0.05 * untimed_rates[-1]:
                break
        for _ in range(chunks):
            results, rate = one_chunk()
            done += sum(r[0] for r in results)
            rates.append(rate)
            busy.extend(r[0] / r[1] for r in results)  # each worker's own rate while it ran
    variants = {}
    # (i): this process with its BLAS pools limited to one thread for the duration of the measurement ONLY
    global _CPU_STATE
    from oracle import ref_like

    _CPU_STATE = (ref_like, ref_like.SpectrumState(sight.y, sight.v, sight.this_mu, sight.this_M, sight.this_omega2,
                                                   sight.padded_wavelengths, sight.unmasked_wavelengths, sight.keep))
    try:
        from threadpoolctl import threadpool_limits

        with threadpool_limits(limits=1):
            rate_i, n_i = _single_process_rate(evals, single_seconds)
        variants["i_one_process_one_blas_thread"] = {"value": rate_i, "cores": 1, "evals": n_i}
    except Exception:
        pass
    try:  # (ii): a fresh interpreter with the default thread count
        spawn = mp.get_context("spawn")
        parent, child = spawn.Pipe(duplex=False)
        proc = spawn.Process(target=_default_threads_child, args=(child, sight, evals, single_seconds, cores))
        # the child inherits THIS process's affinity at its start, and its BLAS sizes its pool when NumPy is imported --
        # which a spawned interpreter does while it re-imports this module, before the target function runs: so the
        # affinity is narrowed to the job's share here, around the start, and restored right after
        before = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
        try:
            if before is not None:
                os.sched_setaffinity(0, set(cores))
            proc.start()
        finally:
            if before is not None:
                os.sched_setaffinity(0, before)
        child.close()
        answer = parent.recv() if parent.poll(120.0) else {"error": "no answer within 120 s"}
        proc.join(30.0)
        if proc.is_alive():
            proc.kill()  # the exact child started above
        if "value" in answer:
            variants["ii_one_process_default_blas_threads"] = {
                "value": answer["value"], "cores": answer["blas_threads_used"], "evals": answer["evals"],
                "blas_threads_used": answer["blas_threads_used"], "affinity_cores": workers}
        else:
            variants["ii_one_process_default_blas_threads"] = answer
    except Exception as exc:  # noqa: BLE001
        variants["ii_one_process_default_blas_threads"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    rates_sorted = sorted(rates)
    return {
        "value": float(np.median(rates)),
        "unit": "evals/s",
        "cores": workers,
        "kind": "port",
        "sample": "%d evals drawn from the %d parameter sets of one step's workload: %d timed chunks of %d evals on each of %d "
                  "pinned processes x 1 BLAS thread after one untimed chunk of the same size; value = median chunk rate"
                  % (done, len(evals), chunks, per_worker, workers),
        "chunk_rates": rates,
        "untimed_ramp_chunk_rates": untimed_rates,
        "spread": (rates_sorted[-1] - rates_sorted[0]) / float(np.median(rates)),
        "per_core": float(np.median(busy)),
        "cores_why": why,
        "variants": variants,
        "host_cpus": os.cpu_count(),
        "host": host_description(),
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


def parse_cpu_list(text: str):
    """Linux cpulist ("0-23,96-119") -> sorted list of ints"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(cpus))


def gpu_local_cpus(index: int, sysfs: str = "/sys", visible=None):
    """
    CPUs of the NUMA node that GPU `index` of this process's device numbering hangs off, read from sysfs WITHOUT touching
    the GPU (a rank binds itself before its first HIP call): the KFD topology lists the compute nodes in the runtime's
    enumeration order (`simd_count` > 0: a GPU; `drm_render_minor` names its DRM node), `ROCR_VISIBLE_DEVICES` and then
    `HIP_VISIBLE_DEVICES` (lists of ordinals; the second indexes into what the first left) select and reorder them, and the DRM device's `local_cpulist` (or the
    `cpulist` of its `numa_node`) gives the CPUs.  Returns (cpus, description); ([], why) when anything is missing --
    binding is an optimisation, never a requirement.
    """
    nodes_dir = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        node_ids = sorted(int(n) for n in os.listdir(nodes_dir) if n.isdigit())
    except OSError:
        return [], "no KFD topology under %s" % nodes_dir
    gpus = []
    for node in node_ids:
        props = {}
        try:
            with open(os.path.join(nodes_dir, str(node), "properties")) as f:
                for ln in f:
                    key, _, value = ln.partition(" ")
                    props[key] = value.strip()
        except OSError:
            continue
        if int(props.get("simd_count", "0") or 0) > 0 and "drm_render_minor" in props:
            gpus.append(int(props["drm_render_minor"]))
    # The two lists COMPOSE: ROCR_VISIBLE_DEVICES filters and reorders what the HSA runtime enumerates, and
    # HIP_VISIBLE_DEVICES then indexes into THAT.  Only lists of ordinals are understood: a UUID ("GPU-...") or anything
    # else that is not an integer makes the answer "unknown" -- no binding -- rather than a guess.
    if visible is None:
        lists = [os.environ.get("ROCR_VISIBLE_DEVICES", ""), os.environ.get("HIP_VISIBLE_DEVICES", "")]
    else:
        lists = [visible]
    for which in lists:
        if not which.strip():
            continue
        try:
            gpus = [gpus[int(tok)] for tok in which.split(",") if tok.strip()]
        except (ValueError, IndexError):
            return [], "cannot interpret the visible-devices list %r" % which
    if not 0 <= index < len(gpus):
        return [], "GPU %d not among the %d GPUs of the KFD topology" % (index, len(gpus))
    device = os.path.join(sysfs, "class", "drm", "renderD%d" % gpus[index], "device")
    try:
        with open(os.path.join(device, "local_cpulist")) as f:
            cpus = parse_cpu_list(f.read())
        if cpus:
            return cpus, "local_cpulist of renderD%d" % gpus[index]
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(device, "numa_node")) as f:
            numa = int(f.read().strip())
        if numa < 0:
            return [], "renderD%d reports no NUMA node" % gpus[index]
        with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % numa, "cpulist")) as f:
            return parse_cpu_list(f.read()), "NUMA node %d of renderD%d" % (numa, gpus[index])
    except (OSError, ValueError):
        return [], "no NUMA information for renderD%d" % gpus[index]


def bind_to_gpu_numa_node(local_rank: int, sysfs: str = "/sys"):
    """--bind: pin this rank (before any GPU call) to the CPUs next to its GPU; returns what was done, for the line"""
    cpus, why = gpu_local_cpus(local_rank, sysfs)
    allowed = set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else set()
    target = sorted(set(cpus) & allowed)
    if not target:
        return {"bound": False, "why": why if not cpus else "none of the GPU's local CPUs is in this process's affinity mask"}
    try:
        os.sched_setaffinity(0, target)
    except OSError as exc:
        return {"bound": False, "why": "sched_setaffinity failed: %s" % exc}
    return {"bound": True, "cpus": len(target), "first_cpu": target[0], "source": why}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--strong", action="store_true", help="config 2: split ONE sightline's samples over the ranks")
    ap.add_argument("--ramp-ms", type=float, default=150.0, help=argparse.SUPPRESS)  # untimed clock ramp before the warmup
    ap.add_argument("--samples", type=int, default=NUM_SAMPLES, help=argparse.SUPPRESS)
    ap.add_argument("--sightlines", type=int, default=0, help=argparse.SUPPRESS)  # tuning: override the batch width
    # unmasked pixels per synthetic sightline (2 % masked): real DR12Q sightlines keep <= 1250, median ~685
    # (reference set_parameters.py:33-34,58; tools/measure_frac_vs_n.py, profiles/r05_frac_vs_n.json)
    ap.add_argument("--pixels", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-other-configs", action="store_true", help=argparse.SUPPRESS)  # headline measurement only
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)  # print that object and leave
    ap.add_argument("--splits", type=int, default=0, help=argparse.SUPPRESS)  # tuning: force the pixel-split count
    ap.add_argument("--option", action="append", default=[], help=argparse.SUPPRESS)  # tuning: library option name=value
    ap.add_argument("--debug", type=int, default=0, help=argparse.SUPPRESS)  # 2: per-phase cycle stamps (diagnostic)
    # rehearsal of the N > 1 code path on a one-GPU box: every rank on device 0, gloo instead of RCCL
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help=argparse.SUPPRESS)
    # every rank pins itself to the CPUs of its GPU's NUMA node before its first GPU call (sysfs only); off by default:
    # the driver's launcher may already have bound the ranks
    ap.add_argument("--bind", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    if args.strong and args.config != 2:
        raise SystemExit("--strong splits the samples of configs[1]'s single sightline: use --config 2")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    return args


def self_launch(args, argv) -> int:
    """
    Plain `python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, as a CHILD
    `python -m torch.distributed.run` (never an exec: this process may not be replaced once anything has touched the
    GPU, and nothing here has -- torch is not imported yet), relay rank 0's one JSON line and return the child's exit
    code.  Under torch.distributed.run (WORLD_SIZE set) this is never reached.
    """
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    try:
        for raw in child.stdout:
            # rank 0's line goes to stdout untouched; anything else a rank printed is kept out of the one-line contract
            (sys.stdout if raw.startswith("{") else sys.stderr).write(raw)
            (sys.stdout if raw.startswith("{") else sys.stderr).flush()
        return child.wait()
    except BaseException:
        child.terminate()  # the exact child we started; torch.distributed.run forwards the signal to its ranks
        try:
            child.wait(timeout=30)
        except subprocess.TimeoutExpired:
            child.kill()
        raise


class Env:
    """what every measurement of this process shares: its rank, its device, the process group"""

    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.rehearse = bool(args.rehearse_on_one_gpu)
        self.local_rank = 0 if self.rehearse else int(os.environ.get("LOCAL_RANK", "0"))
        # launched through torch.distributed.run: the process group and every collective run even with one rank (a
        # one-GPU box then exercises the RCCL calls of the N > 1 path); plain `python bench.py` has no group
        self.dist_on = self.world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
        self.torch = self.dist = self.device = None

    def start(self):
        import torch
        import torch.distributed as dist

        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback)")
        torch.cuda.set_device(self.local_rank)
        self.torch, self.dist = torch, dist
        self.device = torch.device("cuda", self.local_rank)
        if self.dist_on:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.device)

    def all_gather(self, out, mine):
        if self.dist_on and self.rehearse:  # gloo has no all_gather_into_tensor for device tensors
            self.dist.all_gather(list(out.view(self.world, -1).unbind(0)), mine)
        elif self.dist_on:
            self.dist.all_gather_into_tensor(out, mine)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.dist_on:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, values):
        """element-wise maximum of a short list of floats over the ranks (NaN-free inputs)"""
        if not self.dist_on:
            return list(values)
        t = self.torch.tensor(list(values), dtype=self.torch.float64,
                              device="cpu" if self.rehearse else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]


class Workload:
    """
    One configuration's sightlines, sample tables and output slots resident on this rank's GPU, and `step()`: one
    pass of the hot path over them, enqueued on torch's current stream.
    """

    def __init__(self, env, config, S, nb, strong, slots, pixels=0):
        from gpy_dla_detection_amd import sightline as sl
        from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline

        torch = env.torch
        self.env, self.config, self.S, self.nb, self.strong, self.slots, self.sl = env, config, S, nb, strong, slots, sl
        first = 0 if strong else env.rank * nb  # weak scaling: every rank has its own sightlines
        n_u = pixels or N_UNMASKED
        n_masked = N_MASKED if not pixels else max(1, round(pixels * N_MASKED / N_UNMASKED))
        self.sights = [make_sightline(first + i, n_unmasked=n_u, n_masked=n_masked, k=RANK_K) for i in range(nb)]
        self.z_host = np.stack([make_qmc_samples(s, S)[0] for s in self.sights])
        self.nhi_dla_host = make_qmc_samples(self.sights[0], S)[2]
        self.nhi_sub_host = make_qmc_samples(self.sights[0], S, sub_dla=True)[2]
        self.devs = self.batch = None
        self.max_dlas = 4 if config == 5 else 1
        # Every step's evidences land in their own slot; the ranks exchange them with ONE all-gather at the end of
        # the timed region -- the communication pattern of a sharded catalogue run (parallel.model_evidences_sharded).
        self.per_step = {2: 1, 3: 2 * nb, 4: 3 * nb, 5: 4 * nb}[config]
        self.counter = 0
        self.lo, self.hi = 0, S
        if strong:
            per_rank = -(-S // env.world)
            self.lo, self.hi = min(env.rank * per_rank, S), min((env.rank + 1) * per_rank, S)
            self.per_rank = per_rank
        self.evals_per_step = {2: float(self.hi - self.lo), 3: 2.0 * nb * S, 4: nb * (1.0 + 2 * S), 5: 4.0 * nb * S}[config]
        self.mfma_evals_per_step = {2: self.evals_per_step, 3: 2.0 * nb * S, 4: 2.0 * nb * S, 5: 4.0 * nb * S}[config]
        self.flops = algorithmic_flops_per_eval(self.sights[0].n, self.sights[0].k)

    def upload(self):
        env, sl, torch, S, nb = self.env, self.sl, self.env.torch, self.S, self.nb
        f64 = dict(dtype=torch.float64, device=env.device)
        self.devs = [sl.DeviceSightline.from_state(s, device=env.local_rank) for s in self.sights]
        self.batch = sl.DeviceBatch(self.devs) if self.config != 2 else None
        self.z = torch.from_numpy(self.z_host).to(env.device)
        self.nhi_dla = torch.from_numpy(self.nhi_dla_host).to(env.device).contiguous()
        self.nhi_sub = torch.from_numpy(self.nhi_sub_host).to(env.device)
        self.evidences = torch.zeros(self.slots * self.per_step, **f64)
        self.gathered = torch.zeros(env.world * self.slots * self.per_step, **f64) if env.dist_on else None
        if self.config == 2:
            self.out = torch.empty(S, **f64)
            if self.strong:
                self.local = torch.zeros(self.per_rank, **f64)
                self.everything = torch.zeros(env.world * self.per_rank, **f64)
        else:
            from gpy_dla_detection_amd import parallel

            self.uniforms = (torch.from_numpy(parallel.resampling_uniforms(nb, self.max_dlas, S, seed=0)).to(env.device)
                             if self.max_dlas > 1 else None)
            self.sample_ll = torch.empty((nb, self.max_dlas, S), **f64)
            self.sample_ll_sub = torch.empty((nb, 1, S), **f64) if self.config in (3, 4) else None
            self.base = torch.zeros((nb, max(self.max_dlas - 1, 1), S), dtype=torch.int32, device=env.device)
        return self

    def step(self):
        env, sl, torch, S, nb, config = self.env, self.sl, self.env.torch, self.S, self.nb, self.config
        slot = self.counter % self.slots
        self.counter += 1
        stream = torch.cuda.current_stream().cuda_stream
        if config == 2:
            z, nhi = self.z, self.nhi_dla
            if not self.strong:
                # the level's log-likelihoods ("- log S") and its log-mean-exp model evidence (reference dla_gp.py:132-190) in one
                # call: the evidence comes out of the finalisation kernel's tail, no launch of its own
                self.devs[0].eval_evidence_device(S, 1, z.data_ptr(), nhi.data_ptr(), self.out.data_ptr(), 0,
                                                  self.evidences.data_ptr() + 8 * slot, stream=stream)
                return
            # this rank's slice, then all S log-likelihoods on every rank (S / N doubles per rank over RCCL)
            lo, hi = self.lo, self.hi
            self.devs[0].eval_device(hi - lo, 1, z.data_ptr() + 8 * lo, nhi.data_ptr() + 8 * lo, self.local.data_ptr(), stream)
            if env.dist_on:
                env.all_gather(self.everything, self.local)
            else:
                self.everything.copy_(self.local)
            # "- log S" and the log-mean-exp model evidence of the level (reference dla_gp.py:155-190)
            sl.level_evidence_device(self.everything.data_ptr(), 1, S, 0, self.evidences.data_ptr() + 8 * slot,
                                     device=env.local_rank, stream=stream)
            return
        ev = self.evidences.data_ptr() + 8 * slot * self.per_step
        z, batch = self.z.data_ptr(), self.batch
        if config == 5:
            batch.sweep_device(S, 4, z, self.nhi_dla.data_ptr(), 0, MIN_Z_SEPARATION, self.uniforms.data_ptr(), ev,
                               self.sample_ll.data_ptr(), self.base.data_ptr(), stream)
            return
        # the sub-DLA model and DLA(1) sample the same redshifts: one call, and -- the batch's shape allowing -- one launch
        # with one Voigt optical depth per QMC point for both (dlagp_batch_two_model_sweep_device)
        batch.two_model_sweep_device(S, 1, z, self.nhi_sub.data_ptr(), 0, self.nhi_dla.data_ptr(), 0, MIN_Z_SEPARATION, 0,
                                     ev, self.sample_ll_sub.data_ptr(), ev + 8 * nb, self.sample_ll.data_ptr(), 0, stream)
        if config == 4:  # NullGP.log_model_evidence of every member: one launch, host-pointer entry
            self.evidences[slot * self.per_step + 2 * nb: slot * self.per_step + 3 * nb] = \
                torch.from_numpy(batch.nulls()).to(env.device)

    def exchange(self):
        self.env.all_gather(self.gathered, self.evidences)

    def gathered_matches(self) -> bool:
        """every rank holds every rank's evidences after the exchange: rank r's slots must equal its own"""
        torch = self.env.torch
        mine = self.gathered.view(self.env.world, -1)[self.env.rank]
        return bool(torch.equal(torch.nan_to_num(mine), torch.nan_to_num(self.evidences)))

    def total_evals_per_step(self) -> float:
        return float(self.S) if self.strong else self.evals_per_step * self.env.world

    def close(self):
        if self.batch is not None:
            self.batch.close()
        for dev in self.devs or []:
            dev.close()
        self.devs = self.batch = None
        for name in ("z", "nhi_dla", "nhi_sub", "evidences", "gathered", "out", "local", "everything", "uniforms",
                     "sample_ll", "sample_ll_sub", "base"):
            if hasattr(self, name):
                delattr(self, name)


def timed_region(env, work, steps):
    """
    EXACTLY `steps` steps between two fences (synchronize + barrier + synchronize on both sides), the evidences
    exchanged once inside it; returns (elapsed seconds of this rank, accumulate-kernel ms, accumulate launches).
    """
    sl = work.sl
    env.fence()
    work.counter = 0
    sl.set_option("timing", 1)
    sl.timing_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        work.step()
    work.exchange()
    env.fence()
    elapsed = time.perf_counter() - t0
    sl.set_option("timing", 0)
    kernel_ms, launches = sl.timing_read(reset=True)
    return elapsed, kernel_ms, launches


def measure_side_config(env, config, steps, warmup, strong=False):
    """
    One of the other BASELINE configurations (or the by-sample split of configs[1]) at full size, measured after the
    headline: `warmup` untimed steps (the first one builds the launch plan and the workspaces), then `steps` steps
    under the headline's own timing rule.  Every rank reaches every collective whatever happens on another rank: a
    failed set-up is agreed on (max-reduce of a flag) BEFORE any step runs, and reported instead of raised -- the
    headline measurement above is already taken and must not be lost to a side measurement.
    """
    S, nb = NUM_SAMPLES, WORKLOADS[config]["sightlines"]
    work, error = None, None
    try:
        work = Workload(env, config, S, nb, strong, slots=max(steps, warmup, 1)).upload()
        # the device idled while this workload was built: the same untimed clock ramp as in front of the headline
        # (time-based on local work; --strong communicates per step, so there a fixed number of steps)
        if strong and env.dist_on:
            for _ in range(300):
                work.step()
        else:
            ramp_start = time.perf_counter()
            while (time.perf_counter() - ramp_start) * 1e3 < env.args.ramp_ms:
                for _ in range(10 if config == 2 else 1):
                    work.step()
                env.torch.cuda.synchronize()
        for _ in range(warmup):
            work.step()
        env.torch.cuda.synchronize()
    except Exception as exc:  # noqa: BLE001 -- reported in the line
        error = "%s: %s" % (type(exc).__name__, exc)
    failed = env.max_over_ranks([1.0 if error else 0.0])[0] > 0
    result = {"workload": "configs[%d]%s: %s" % (WORKLOADS[config]["baseline_index"], " --strong" if strong else "",
                                                "S/N samples of the one sightline per rank, one all-gather of the "
                                                "log-likelihoods per step" if strong else WORKLOADS[config]["what"]),
              "n_gpus": env.world, "steps": steps, "warmup": warmup}
    if failed:
        result["error"] = error or "set-up failed on another rank"
    else:
        elapsed, kernel_ms, launches = timed_region(env, work, steps)
        elapsed, kernel_s = env.max_over_ranks([elapsed, kernel_ms / 1e3])
        value = work.total_evals_per_step() * steps / elapsed
        achieved = work.mfma_evals_per_step * steps * work.flops / kernel_s / 1e12 if kernel_s > 0 else 0.0
        result.update({
            "ms_per_step": elapsed / steps * 1e3,
            "evals_per_s": value,
            "spectra_per_s": nb * env.world * steps / elapsed if config in (4, 5) else None,
            "sightlines_per_gpu": nb if not strong else 1.0 / env.world,
            "scaling": "strong" if strong else "weak",
            "accumulate_ms_per_step": kernel_s * 1e3 / steps,
            "accumulate_launches": launches,
            "accumulate_tflops": achieved,
            "accumulate_frac": achieved / PEAK_FP64_MFMA_TFLOPS,
            "evidence": float(work.evidences[0].item()),
        })
        if env.dist_on and not work.gathered_matches():
            result["error"] = "rank %d: the gathered evidences differ from the local ones" % env.rank
    if work is not None:
        work.close()
    del work
    env.torch.cuda.empty_cache()
    return result


def main(argv=None) -> None:
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, argv))
    config = args.config
    steps = args.steps if args.steps is not None else (200 if config == 2 else 5)
    warmup = args.warmup if args.warmup is not None else (20 if config == 2 else 1)

    env = Env(args)
    binding = bind_to_gpu_numa_node(env.local_rank) if args.bind else None
    if env.world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, env.world))
    world, rank = env.world, env.rank

    S = args.samples
    nb = args.sightlines or WORKLOADS[config]["sightlines"]
    work = Workload(env, config, S, nb, args.strong, slots=max(steps, warmup, 1), pixels=args.pixels)
    sights = work.sights

    # the CPU baseline forks worker processes: do it before this process touches the GPU
    baseline = None
    if world == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        baseline = cpu_baseline(sights[0], baseline_sample(config, sights[0], work.z_host[0], work.nhi_dla_host,
                                                           work.nhi_sub_host))

    if args.cpu_baseline_only:
        print(json.dumps(baseline), flush=True)
        return
    env.start()
    torch, dist, sl = env.torch, env.dist, work.sl
    work.upload()
    if args.splits:
        sl.set_option("pixel_splits", args.splits)
    if args.debug:
        sl.set_option("debug", args.debug)
    for item in args.option:
        name, value = item.split("=")
        sl.set_option(name, int(value))
    step = work.step

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

    flops = work.flops

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
    if args.strong and env.dist_on:
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
    work.exchange()
    elapsed, kernel_ms, launches = timed_region(env, work, steps)
    if args.debug & 2:
        st = sl.debug_read()
        if rank == 0 and st[5]:
            names = ["voigt", "conv", "mfma_loop", "epilogue", "passes+epilogue", None, "prologue"]
            print("stamps: waves=%d  " % st[5] + "  ".join("%s=%.0f" % (n, st[i] / st[5]) for i, n in enumerate(names) if n) + "  (cycles per wave)", file=sys.stderr)

    rank_spread = None
    if env.dist_on:
        # the slowest rank's time is the job's; the fastest rank's beside it shows an imbalance without a second run
        slowest, minus_fastest = env.max_over_ranks([elapsed, -elapsed])
        rank_spread = {"min": -minus_fastest / steps * 1e3, "max": slowest / steps * 1e3}
        elapsed = slowest
        if not work.gathered_matches():
            raise SystemExit("rank %d: the gathered evidences differ from the local ones" % rank)
    evidence = float(work.evidences[0].item())
    evals_per_step, mfma_evals_per_step = work.evals_per_step, work.mfma_evals_per_step
    total_evals = work.total_evals_per_step() * steps
    n_pixels, rank_k = sights[0].n, sights[0].k
    work.close()

    # The other BASELINE configurations at their one-GPU widths and the by-sample split of configs[1], measured after
    # the headline (whose figures above are final by now) so that the driver's one run carries them too: clock ramp, one
    # warm-up step and three timed steps each (config 3: 36 ms, config 4: 73 ms, config 5: 38 ms per step).
    others = None
    if (config == 2 and not args.strong and not args.no_other_configs and S == NUM_SAMPLES and not args.sightlines
            and not args.splits and not args.option and not args.debug and not args.pixels):
        others = {str(c): measure_side_config(env, c, steps=3, warmup=1) for c in (3, 4, 5)}
        strong = measure_side_config(env, 2, steps=100, warmup=20, strong=True)

    if rank == 0:
        value = total_evals / elapsed
        kernel_s_total = kernel_ms / 1e3
        # timed accumulate launches of this rank: all MFMA-path launches of the step (config 4's null launch is the
        # generic kernel on one sample per sightline and is timed with them; its FLOPs are not counted)
        achieved = mfma_evals_per_step * steps * flops / kernel_s_total / 1e12 if kernel_s_total > 0 else 0.0
        traffic, traffic_file = measured_hbm_traffic(config, n_pixels, S)
        # the whole step against the same peak: everything between two steps' starts -- finalisation, level evidence,
        # launch boundaries -- counts, which is what `value` corresponds to (`frac` is the dominant kernel alone)
        step_achieved = mfma_evals_per_step * steps * flops / elapsed / 1e12
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
                            % (WORKLOADS[config]["baseline_index"], config, WORKLOADS[config]["what"], S, n_pixels,
                               args.pixels or N_UNMASKED, rank_k),
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
                "step_achieved": step_achieved,
                "step_frac": step_achieved / PEAK_FP64_MFMA_TFLOPS,
                "traffic": traffic,
                "traffic_unit": "HBM bytes per accumulate launch (PMC, profiles/%s)" % traffic_file if traffic_file else None,
                "kernel": "accumulate_mfma (all instantiations launched in the timed region)",
                "kernel_ms": kernel_ms / max(launches, 1),
                "kernel_launches": launches,
                "flop_per_eval": flops,
            },
            "evidence": evidence,
        }
        if rank_spread is not None:
            line["ms_per_step_ranks"] = rank_spread
        if binding is not None:
            line["cpu_binding_rank0"] = binding
        if others is not None:
            line["other_configs"] = others
            line["strong_n%d" % world] = strong
        if baseline is not None:
            line["cpu_baseline"] = baseline
        print(json.dumps(line), flush=True)

    if env.dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
