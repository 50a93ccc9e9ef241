"""
Batches of sightlines on one GPU, and sightlines sharded over the GPUs of a node.

The reference processes one spectrum at a time in a single Python process
(run_bayes_select.py:141-245).  Spectra are independent, so this module
  * evaluates a whole batch of sightlines per occupation level with one launch sequence
    (`sweep_batch`: DeviceBatch, samples laid out [sightline][sample][absorber]); the host half of
    every level (separation mask, evidence, `np.random.choice` resampling) runs per sightline
    with the reference's seeding (`np.random.seed(0)` per spectrum, run_bayes_select.py:144) by
    saving and restoring the legacy global RNG state around each sightline's turn;
  * shards sightlines over ranks in contiguous blocks (one process per GPU, no data-path
    collective) and gathers the per-sightline log-evidences to every rank with one all-gather --
    RCCL over xGMI when the process group is `nccl`, `gloo` in the CPU tests.

Message sizes: (2 + max_dlas) doubles per sightline -- 48 B for the default model list
[null, sub-DLA, DLA(1..4)] -- so the gather is latency-bound (1024 sightlines = 48 KB).
"""
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .dla_gp import finish_level, level_parameters


def shard_range(num_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """contiguous block [lo, hi) of `rank`; the first `num_items % world_size` ranks get one extra"""
    base, extra = divmod(num_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_rows(local_rows: np.ndarray, total_rows: int, group=None, device=None) -> np.ndarray:
    """
    All-gather row blocks distributed by `shard_range` into the full (total_rows, m) array on every
    rank.  Blocks are padded to the largest shard so that one fixed-size collective suffices.
    Without an initialised process group this is the identity (single process).
    """
    import torch
    import torch.distributed as dist

    local_rows = np.ascontiguousarray(local_rows, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        assert local_rows.shape[0] == total_rows
        return local_rows
    world = dist.get_world_size(group)
    m = local_rows.shape[1]
    largest = shard_range(total_rows, world, 0)[1]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    send = torch.full((largest, m), float("nan"), dtype=torch.float64, device=device)
    send[: local_rows.shape[0]] = torch.from_numpy(local_rows).to(device)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    out = np.empty((total_rows, m), dtype=np.float64)
    for r in range(world):
        lo, hi = shard_range(total_rows, world, r)
        out[lo:hi] = recv[r][: hi - lo].cpu().numpy()
    return out


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
    outer_state = np.random.get_state()
    np.random.seed(seed)
    one = np.random.random_sample((levels, S))
    np.random.set_state(outer_state)
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
) -> np.ndarray:
    """
    log-evidences [null, sub-DLA, DLA(1..max_dlas)] of every sightline in `sightlines` (objects with
    the post-set_data attributes, e.g. synthetic.SyntheticSightline), computed by the rank that owns
    the sightline and all-gathered: returns (len(sightlines), 2 + max_dlas) on every rank.

    `make_batch(list of sightlines, device)` must return an object with `.eval(z, nhi)` and
    `.nulls()` (and optionally `.sweep(...)`); the default builds DeviceSightline / DeviceBatch (GPU).
    """
    total = len(sightlines)
    lo, hi = shard_range(total, world_size, rank)
    runner = ShardRunner(sightlines[lo:hi], num_samples, max_dlas, min_z_separation_kms, device, make_batch)
    return gather_rows(runner.run(), total, group=group)


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
