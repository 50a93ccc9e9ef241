"""
Batches of sightlines on one GPU, and sightlines sharded over the GPUs of a node.

The reference processes one spectrum at a time in a single Python process
(run_bayes_select.py:141-245).  Spectra are independent, so this module
  * evaluates a whole batch of sightlines per occupation level with one launch sequence
    (`sweep_batch`: DeviceBatch, samples laid out [sightline][sample][absorber]); the host half of
    every level (separation mask, evidence, `np.random.choice` resampling) runs per sightline
    with the reference's seeding (`np.random.seed(0)` per spectrum, run_bayes_select.py:144) by
    saving and restoring the legacy global RNG state around each sightline's turn;
  * shards sightlines over ranks (one process per GPU, no data-path collective) by COST, not by count: the work of
    a spectrum goes with its kept pixels n, which follow z_qso through the modelled window (reference
    null_gp.py:139-152, set_parameters.py:125-159: ~200 pixels at z = 2.15, 1250 from z = 3), so equal counts of a
    redshift-ordered catalogue leave the low-z ranks idle.  `balanced_assignment` is a deterministic
    longest-processing-time assignment every rank computes identically from the inputs; the per-sightline
    log-evidences return to every rank, in catalogue order, with one padded all-gather -- RCCL over xGMI when the
    process group is `nccl`, `gloo` in the CPU tests;
  * splits the samples of ONE sightline over the ranks (`sweep_sharded_samples*`, the single-spectrum
    scaling point): the S samples of an occupation level are independent (reference dla_gp.py:132-159;
    MATLAB parallelises this axis, multi_dlas/process_qsos_multiple_dlas_meanflux.m:333), the
    resampling between levels needs all of them (:209-218) -- one all-gather of S/G doubles per level,
    then every rank finishes the level with the same random numbers, so `base_sample_inds` agree on all
    ranks without a broadcast.

Message sizes: (2 + max_dlas) doubles per sightline -- 48 B for the default model list
[null, sub-DLA, DLA(1..4)] -- so the gather is latency-bound (1024 sightlines = 48 KB).
"""
import threading
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .dla_gp import finish_level, level_parameters


# `sweep_batch` re-seeds and restores the legacy global NumPy RNG around every sightline's turn (the reference resamples
# with `np.random.choice` on that generator): one sweep at a time per process
_GLOBAL_RNG_LOCK = threading.RLock()


def shard_range(num_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """contiguous block [lo, hi) of `rank`; the first `num_items % world_size` ranks get one extra"""
    base, extra = divmod(num_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def spectrum_cost(n_pixels, num_samples: int, max_dlas: int, sub_dla: bool = True):
    """
    Modelled cost of the model-evidence sweeps of one spectrum, in pixel-evaluations: n kept pixels x (1 null
    evaluation + S samples x (sub-DLA model + max_dlas occupation levels)).  Every term of an evaluation is linear
    in n (Voigt profile, weights, the n k (k + 1) contraction: SURVEY.md 8d).  Levels a sweep skips after a NaN
    evidence (reference dla_gp.py:200-206) are not known in advance and are charged in full.
    """
    n = np.asarray(n_pixels, dtype=np.float64)
    return n * (1.0 + float(num_samples) * ((1.0 if sub_dla else 0.0) + float(max_dlas)))


def balanced_assignment(costs, world_size: int) -> List[np.ndarray]:
    """
    Longest-processing-time assignment of items to ranks: items in order of decreasing cost (ties: catalogue order),
    each to the rank with the least cost so far (ties: lowest rank).  Deterministic, so every rank computes the same
    table from the same inputs and no broadcast is needed.  Returns, per rank, its item indices in catalogue order.
    Greedy LPT is within 4/3 - 1/(3 ranks) of the optimum in the worst case and within a fraction of the smallest item
    for catalogues of many more items than ranks.
    """
    import heapq

    costs = np.asarray(costs, dtype=np.float64)
    if costs.ndim != 1 or world_size < 1:
        raise ValueError("costs must be 1-d and world_size >= 1")
    if not np.all(np.isfinite(costs)) or np.any(costs < 0):
        raise ValueError("costs must be finite and non-negative")
    order = np.lexsort((np.arange(costs.shape[0]), -costs))
    heap = [(0.0, r) for r in range(world_size)]
    owner = np.empty(costs.shape[0], dtype=np.int64)
    for item in order:
        load, r = heapq.heappop(heap)
        owner[item] = r
        heapq.heappush(heap, (load + float(costs[item]), r))
    return [np.flatnonzero(owner == r) for r in range(world_size)]


def contiguous_assignment(num_items: int, world_size: int) -> List[np.ndarray]:
    """`shard_range` as an assignment table (equal counts of consecutive items)"""
    return [np.arange(*shard_range(num_items, world_size, r)) for r in range(world_size)]


def assignment_imbalance(costs, assignment: Sequence[np.ndarray]) -> float:
    """busiest rank's cost / mean cost per rank (1.0 = perfect balance)"""
    costs = np.asarray(costs, dtype=np.float64)
    loads = np.array([costs[idx].sum() for idx in assignment])
    return float(loads.max() / loads.mean()) if loads.mean() > 0 else 1.0


def gather_rows(local_rows: np.ndarray, total_rows: int, group=None, device=None,
                assignment: Optional[Sequence[np.ndarray]] = None) -> np.ndarray:
    """
    All-gather the row blocks of the ranks into the full (total_rows, m) array, in catalogue order, on every rank.
    `assignment[r]` = the rows rank r holds, in the order it holds them (default: `shard_range` blocks).  Blocks
    are padded to the largest shard so that one fixed-size collective suffices.
    Without an initialised process group this is the identity (single process).
    """
    import torch
    import torch.distributed as dist

    # the rows travel in their own type: base indices as int32 (index data has no business in floating point), everything
    # else as float64
    local_rows = np.asarray(local_rows)
    dtype = local_rows.dtype if local_rows.dtype in (np.int32, np.int64) else np.dtype(np.float64)
    local_rows = np.ascontiguousarray(local_rows, dtype=dtype)
    if not (dist.is_available() and dist.is_initialized()):
        assert local_rows.shape[0] == total_rows
        if assignment is not None:
            out = np.empty_like(local_rows)
            out[np.asarray(assignment[0])] = local_rows
            return out
        return local_rows
    world = dist.get_world_size(group)
    if assignment is None:
        assignment = contiguous_assignment(total_rows, world)
    if len(assignment) != world or sum(len(idx) for idx in assignment) != total_rows:
        raise ValueError("assignment does not cover %d rows over %d ranks" % (total_rows, world))
    if local_rows.shape[0] != len(assignment[dist.get_rank(group)]):
        raise ValueError("this rank holds %d rows, its assignment lists %d"
                         % (local_rows.shape[0], len(assignment[dist.get_rank(group)])))
    m = local_rows.shape[1]
    largest = max(len(idx) for idx in assignment)
    if device is None:
        if dist.get_backend(group) == "nccl":
            _require_torch_gpu(torch)
            device = torch.device("cuda", torch.cuda.current_device())
        else:
            device = torch.device("cpu")
    if dtype == np.float64:
        send = torch.full((largest, m), float("nan"), dtype=torch.float64, device=device)
    else:
        send = torch.zeros((largest, m), dtype=torch.int32 if dtype == np.int32 else torch.int64, device=device)
    send[: local_rows.shape[0]] = torch.from_numpy(local_rows).to(device)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    out = np.empty((total_rows, m), dtype=dtype)
    for r in range(world):
        idx = np.asarray(assignment[r])
        out[idx] = recv[r][: len(idx)].cpu().numpy()
    return out


def _require_torch_gpu(torch) -> None:
    """RCCL runs inside torch's HIP runtime, which must be the one libdlagp_hip.so is bound to (`_lib.load`)"""
    from . import _lib

    if not torch.cuda.is_available():
        raise RuntimeError("the process group is `nccl` but torch sees no GPU")
    copies = _lib.hip_runtime_copies()
    if len(copies) > 1:
        raise RuntimeError(
            "two HIP runtimes are mapped into this process (%s): launches of libdlagp_hip.so and RCCL collectives "
            "would run on unrelated streams.  Load the library through gpy_dla_detection_amd._lib.load() without "
            "DLAGP_HIP_RUNTIME=system so that it binds to torch's runtime" % ", ".join(copies))


def sweep_sharded_samples(level_eval: Callable[[np.ndarray, np.ndarray], np.ndarray], sample_z: np.ndarray,
                          sample_nhi: np.ndarray, max_dlas: int, min_z_separation: float,
                          rank: int = 0, world_size: int = 1, group=None):
    """
    Occupation sweep (reference dla_gp.py:92-225) of ONE sightline with the samples of every level split over the
    ranks: rank r evaluates samples `shard_range(S, world_size, r)` (`level_eval(z (s, kd), nhi (s, kd)) -> (s,)`,
    e.g. `DeviceSightline.eval`), the slices are all-gathered (S / world_size doubles per rank and level), and
    every rank runs the level's host half (`dla_gp.finish_level`) on the full vector.  The legacy global NumPy RNG
    must be in the same state on every rank (seed it the same way, as the reference driver does per spectrum,
    run_bayes_select.py:144): the resampled `base_sample_inds` are then identical everywhere and equal to a
    single-process sweep's.

    :return: evidences (max_dlas,), sample_log_likelihoods (S, max_dlas), base_sample_inds (max_dlas - 1, S) int32
    """
    S = sample_z.shape[0]
    lo, hi = shard_range(S, world_size, rank)
    evidences = np.full((max_dlas,), np.nan)
    base_sample_inds = np.zeros((max_dlas - 1, S), dtype=np.int32)
    sample_log_likelihoods = np.full((S, max_dlas), np.nan)
    for level in range(max_dlas):
        z, nhi, idx = level_parameters(sample_z, sample_nhi, base_sample_inds, level)
        local = np.asarray(level_eval(z[lo:hi], nhi[lo:hi]), dtype=np.float64).reshape(hi - lo, 1)
        full = gather_rows(local, S, group=group)[:, 0] if world_size > 1 else local[:, 0]
        sample_log_likelihoods[:, level] = full - np.log(S)  # dla_gp.py:155-159
        evidences[level], keep_going = finish_level(
            sample_log_likelihoods, level, sample_z, idx, min_z_separation, max_dlas, base_sample_inds)
        if not keep_going:
            break
    return evidences, sample_log_likelihoods, base_sample_inds


def sweep_sharded_samples_device(batch, sample_z: np.ndarray, sample_nhi: np.ndarray, max_dlas: int,
                                 min_z_separation: float, uniforms: Optional[np.ndarray] = None,
                                 rank: int = 0, world_size: int = 1, group=None, slices: Optional[Sequence] = None):
    """
    The same by-sample split with everything resident in HBM (C ABI `dlagp_batch_level_eval_device`,
    `dlagp_level_finish_device`): per level one slice evaluation, one all-gather of the slice
    (`torch.distributed.all_gather_into_tensor` = RCCL over xGMI when the group is `nccl`) and the level's
    device half on every rank with the same `uniforms` (`resampling_uniforms`) -- no host round trip.

    :param batch: `sightline.DeviceBatch` of the nb sightlines whose samples are split (normally one)
    :param sample_z: (nb, S); sample_nhi: (S,) shared or (nb, S); uniforms: (nb, max_dlas - 1, S)
    :param slices: testing aid on one GPU: evaluate these [(lo, hi), ...] slices one after the other in this
        process instead of one slice per rank (no process group involved)
    :return: evidences (nb, max_dlas), sample_log_likelihoods (nb, S, max_dlas), base_sample_inds (nb, max_dlas - 1, S)
    """
    import torch
    import torch.distributed as dist

    from . import _lib

    lib = _lib.load()
    nb, S = sample_z.shape
    if max_dlas > 1 and (uniforms is None or uniforms.shape != (nb, max_dlas - 1, S)):
        raise ValueError("uniforms must be (nb, max_dlas - 1, S)")
    # world_size == 1 with no group passed is the LOCAL sweep, also inside a multi-rank job (e.g. one rank of a
    # by-spectrum sharded run sweeping a sightline of its own): no collective.  The collective path runs for
    # world_size > 1, and -- a one-GPU rehearsal of the RCCL calls -- for a single rank when a group is passed
    # explicitly or the default group itself has one rank.
    group_up = dist.is_available() and dist.is_initialized()
    if world_size > 1 and not group_up and slices is None:
        raise RuntimeError("world_size = %d needs an initialised torch.distributed process group" % world_size)
    distributed = group_up and slices is None and (
        world_size > 1 or group is not None or dist.get_world_size() == 1)
    if distributed:
        if dist.get_world_size(group) != world_size or dist.get_rank(group) != rank:
            raise ValueError("rank / world_size do not match the process group")
        if dist.get_backend(group) == "nccl":
            _require_torch_gpu(torch)
    dev = torch.device("cuda", batch.sightlines[0].device)
    f64 = dict(dtype=torch.float64, device=dev)
    d_z = torch.from_numpy(np.ascontiguousarray(sample_z, dtype=np.float64)).to(dev)
    d_nhi = torch.from_numpy(np.ascontiguousarray(sample_nhi, dtype=np.float64)).to(dev)
    nhi_stride = 0 if sample_nhi.ndim == 1 else S
    d_uniforms = torch.from_numpy(np.ascontiguousarray(uniforms, dtype=np.float64)).to(dev) if max_dlas > 1 else None
    d_sample_ll = torch.empty((nb, max_dlas, S), **f64)
    d_evidences = torch.empty((nb, max_dlas), **f64)
    d_base = torch.zeros((nb, max(max_dlas - 1, 1), S), dtype=torch.int32, device=dev)
    levels = max(max_dlas - 1, 0)
    d_flags = torch.zeros(nb + 2 + nb * levels, dtype=torch.int32, device=dev)  # alive, status x 2, near-tie counts
    d_cdf = torch.empty(nb * (S + _lib.LEVEL_EXCHANGE), **f64)
    per_rank = -(-S // world_size)  # equal slices (the last may be short): one fixed-size collective per level
    mine = slices if slices is not None else [(min(rank * per_rank, S), min((rank + 1) * per_rank, S))]
    d_local = torch.zeros((nb, per_rank if slices is None else max(hi - lo for lo, hi in mine)), **f64)
    d_all = torch.zeros((world_size, nb, per_rank), **f64) if distributed else None
    d_ll = torch.empty((nb, S), **f64)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731

    _lib.check(lib.dlagp_sweep_state_init_device(nb, S, max_dlas, ptr(d_sample_ll), ptr(d_evidences), ptr(d_base),
                                                 ptr(d_flags), dev.index, stream))
    for level in range(max_dlas):
        for lo, hi in mine:
            out = d_local[:, : hi - lo].contiguous() if d_local.shape[1] != hi - lo else d_local
            _lib.check(lib.dlagp_batch_level_eval_device(
                batch._handle, S, lo, hi - lo, level, ptr(d_z), ptr(d_nhi), nhi_stride, ptr(d_base),
                max(max_dlas - 1, 0), ptr(out), stream))
            if not distributed:
                d_ll[:, lo:hi] = out
            elif out is not d_local:
                d_local[:, : hi - lo] = out
        if distributed:
            if dist.get_backend(group) == "nccl":
                dist.all_gather_into_tensor(d_all, d_local, group=group)
            else:  # gloo rehearsal
                dist.all_gather(list(d_all.unbind(0)), d_local, group=group)
            d_ll.copy_(d_all.permute(1, 0, 2).reshape(nb, world_size * per_rank)[:, :S])
        _lib.check(lib.dlagp_level_finish_device(
            nb, S, level, max_dlas, ptr(d_ll), ptr(d_z), float(min_z_separation), ptr(d_uniforms), ptr(d_sample_ll),
            ptr(d_evidences), ptr(d_base), ptr(d_cdf), ptr(d_flags), dev.index, stream))
    torch.cuda.synchronize(dev)
    base = d_base[:, : max_dlas - 1].cpu().numpy() if max_dlas > 1 else np.zeros((nb, 0, S), dtype=np.int32)
    evidences = d_evidences.cpu().numpy()
    sample_ll = np.ascontiguousarray(d_sample_ll.cpu().numpy().transpose(0, 2, 1))
    if levels:
        # every rank holds the same likelihoods, uniforms and near-tie counts, so every rank settles the same levels the
        # same way (a level whose indices change is redone locally, all samples on this rank's device: no collective)
        from .sightline import settle_near_ties

        ties = d_flags[nb + 2:].cpu().numpy().reshape(nb, levels)
        if ties.any():
            base = np.ascontiguousarray(base)
            settle_near_ties(ties, evidences, sample_ll, base, np.asarray(uniforms, dtype=np.float64),
                             np.asarray(sample_z, dtype=np.float64), np.asarray(sample_nhi, dtype=np.float64),
                             float(min_z_separation), lambda b, z, n: batch.sightlines[b].eval(z, n))
    return evidences, sample_ll, base


def sweep_batch(
    batch_eval: Callable[[np.ndarray, np.ndarray], np.ndarray],
    sample_z: np.ndarray,
    sample_nhi: np.ndarray,
    max_dlas: int,
    min_z_separation: float,
    seed: Optional[int] = 0,
):
    """
    Occupation sweep (reference dla_gp.py:92-225) of `nb` sightlines at once.

    :param batch_eval: (z (nb, S, kd), nhi (nb, S, kd)) -> (nb, S) log-likelihoods, e.g. DeviceBatch.eval
    :param sample_z: (nb, S) absorber redshifts of every sightline's QMC samples
    :param sample_nhi: (S,) column densities (shared, as in the reference's sample files)
    :param seed: each sightline resamples as if it had been processed alone after `np.random.seed(seed)`
        (the reference driver's behaviour); None leaves the global RNG untouched between sightlines
    :return: evidences (nb, max_dlas), sample_log_likelihoods (nb, S, max_dlas), base_sample_inds (nb, max_dlas - 1, S)
    """
    with _GLOBAL_RNG_LOCK:
        return _sweep_batch_locked(batch_eval, sample_z, sample_nhi, max_dlas, min_z_separation, seed)


def _sweep_batch_locked(batch_eval, sample_z, sample_nhi, max_dlas, min_z_separation, seed):
    nb, S = sample_z.shape
    evidences = np.full((nb, max_dlas), np.nan)
    sample_ll = np.full((nb, S, max_dlas), np.nan)
    base_inds = np.zeros((nb, max_dlas - 1, S), dtype=np.int32)
    alive = np.ones(nb, dtype=bool)
    rng_states = [None] * nb
    if seed is not None:
        outer_state = np.random.get_state()
        np.random.seed(seed)
        rng_states = [np.random.get_state()] * nb

    for level in range(max_dlas):
        z = np.empty((nb, S, level + 1))
        nhi = np.empty((nb, S, level + 1))
        idx = [None] * nb
        for b in range(nb):
            z[b], nhi[b], idx[b] = level_parameters(sample_z[b], sample_nhi, base_inds[b], level)
        ll = batch_eval(z, nhi)
        for b in range(nb):
            if not alive[b]:
                continue
            sample_ll[b, :, level] = ll[b] - np.log(S)
            if seed is not None:
                np.random.set_state(rng_states[b])
            evidences[b, level], alive[b] = finish_level(
                sample_ll[b], level, sample_z[b], idx[b], min_z_separation, max_dlas, base_inds[b]
            )
            if seed is not None:
                rng_states[b] = np.random.get_state()
        if not alive.any():
            break
    if seed is not None:
        np.random.set_state(outer_state)
    return evidences, sample_ll, base_inds


def resampling_uniforms(nb: int, max_dlas: int, S: int, seed: Optional[int] = 0) -> np.ndarray:
    """
    (nb, max_dlas - 1, S) uniforms that make the device resampling draw the reference's indices:
    legacy `np.random.choice(S, S, p=p)` is `searchsorted(cumsum(p) / cumsum(p)[-1], random_sample(S),
    side="right")`, one `random_sample(S)` per level.  With `seed` every sightline gets the stream it
    would see if processed alone after `np.random.seed(seed)` (run_bayes_select.py:144) and the global
    RNG is left as found; with None the draws advance the global RNG sightline by sightline.
    """
    levels = max(max_dlas - 1, 0)
    if seed is None:
        return np.random.random_sample((nb, levels, S))
    # a generator of its own, seeded like the global one would be: the same MT19937 stream as `np.random.seed(seed)`
    # followed by `np.random.random_sample`, without touching the global state (two threads of the catalogue driver
    # draw their batches' uniforms at the same time)
    one = np.random.RandomState(seed).random_sample((levels, S))
    return np.broadcast_to(one, (nb, levels, S)).copy()


def sweep_batch_device(batch, sample_z: np.ndarray, sample_nhi: np.ndarray, max_dlas: int,
                       min_z_separation: float, seed: Optional[int] = 0):
    """
    `sweep_batch` with every level's host half moved to the device (`DeviceBatch.sweep`,
    C ABI dlagp_batch_sweep): same arguments, same return values, one call into the library.
    """
    nb, S = sample_z.shape
    return batch.sweep(sample_z, sample_nhi, max_dlas, min_z_separation,
                       resampling_uniforms(nb, max_dlas, S, seed))


class ShardRunner:
    """
    One rank's share of a catalogue: its sightlines resident on one GPU (contexts, batch, workspace
    are built once and reused), the synthetic Halton QMC sets of SURVEY.md §8(d), and `run()` =
    log-evidences [null, sub-DLA, DLA(1..max_dlas)] of every local sightline.
    """

    def __init__(self, sightlines: Sequence, num_samples: int, max_dlas: int = 4,
                 min_z_separation_kms: float = 3000.0, device: int = 0, make_batch: Optional[Callable] = None):
        from .synthetic import make_qmc_samples

        self.sightlines = list(sightlines)
        self.max_dlas = max_dlas
        self.min_sep = min_z_separation_kms * 1000.0 / 299792458.0
        self.batch = None
        if self.sightlines:
            self.batch = (make_batch or _device_batch)(self.sightlines, device)
            self.z = np.stack([make_qmc_samples(s, num_samples)[0] for s in self.sightlines])
            self.nhi_dla = make_qmc_samples(self.sightlines[0], num_samples)[2]
            self.nhi_sub = make_qmc_samples(self.sightlines[0], num_samples, sub_dla=True)[2]

    def run(self) -> np.ndarray:
        local = np.full((len(self.sightlines), 2 + self.max_dlas), np.nan)
        if self.batch is not None:
            if hasattr(self.batch, "model_evidences"):  # the three models in one call into the library
                S = self.z.shape[1]
                nulls, sub, dla = self.batch.model_evidences(
                    self.z, self.nhi_sub, self.z, self.nhi_dla, self.max_dlas, self.min_sep,
                    resampling_uniforms(len(self.sightlines), self.max_dlas, S, 0))
                local[:, 0], local[:, 1], local[:, 2:] = nulls, sub[0][:, 0], dla[0]
                return local
            local[:, 0] = self.batch.nulls()
            if hasattr(self.batch, "sweep"):  # levels chained on the device
                local[:, 1] = sweep_batch_device(self.batch, self.z, self.nhi_sub, 1, self.min_sep)[0][:, 0]
                local[:, 2:] = sweep_batch_device(self.batch, self.z, self.nhi_dla, self.max_dlas, self.min_sep)[0]
            else:  # an evaluator that only offers eval(): host half per level
                local[:, 1] = sweep_batch(self.batch.eval, self.z, self.nhi_sub, 1, self.min_sep)[0][:, 0]
                local[:, 2:] = sweep_batch(self.batch.eval, self.z, self.nhi_dla, self.max_dlas, self.min_sep)[0]
        return local


def model_evidences_sharded(
    sightlines: Sequence,
    num_samples: int,
    max_dlas: int = 4,
    min_z_separation_kms: float = 3000.0,
    device: int = 0,
    rank: int = 0,
    world_size: int = 1,
    group=None,
    make_batch: Optional[Callable] = None,
    partition: str = "cost",
) -> np.ndarray:
    """
    log-evidences [null, sub-DLA, DLA(1..max_dlas)] of every sightline in `sightlines` (objects with
    the post-set_data attributes, e.g. synthetic.SyntheticSightline), computed by the rank that owns
    the sightline and all-gathered: returns (len(sightlines), 2 + max_dlas) on every rank, rows in the order of
    `sightlines`.  `partition`: "cost" (default) balances the ranks' modelled work (`balanced_assignment` on
    `spectrum_cost` of the kept pixels), "contiguous" cuts equal counts of consecutive sightlines.

    `make_batch(list of sightlines, device)` must return an object with `.eval(z, nhi)` and
    `.nulls()` (and optionally `.sweep(...)`); the default builds DeviceSightline / DeviceBatch (GPU).
    """
    total = len(sightlines)
    if partition == "cost":
        costs = spectrum_cost([s.n for s in sightlines], num_samples, max_dlas)
        assignment = balanced_assignment(costs, world_size)
    elif partition == "contiguous":
        assignment = contiguous_assignment(total, world_size)
    else:
        raise ValueError("partition must be 'cost' or 'contiguous'")
    mine = [sightlines[i] for i in assignment[rank]]
    runner = ShardRunner(mine, num_samples, max_dlas, min_z_separation_kms, device, make_batch)
    return gather_rows(runner.run(), total, group=group, assignment=assignment)


class _DeviceBatchAdapter:
    def __init__(self, sights, device):
        from .sightline import DeviceBatch, DeviceSightline

        self.devs = [DeviceSightline.from_state(s, device=device) for s in sights]
        self.batch = DeviceBatch(self.devs)

    def eval(self, z, nhi):
        return self.batch.eval(z, nhi)

    def sweep(self, *args):
        return self.batch.sweep(*args)

    def nulls(self):
        return self.batch.nulls()


def _device_batch(sights, device):
    return _DeviceBatchAdapter(sights, device)
