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

    The QMC samples are the synthetic Halton sets of SURVEY.md §8(d) (`synthetic.make_qmc_samples`).
    `make_batch(list of sightlines, device)` must return an object with `.eval(z, nhi)` and
    `.nulls()`; the default builds DeviceSightline / DeviceBatch (GPU).
    """
    from .synthetic import make_qmc_samples

    total = len(sightlines)
    lo, hi = shard_range(total, world_size, rank)
    mine = list(sightlines[lo:hi])
    local = np.full((len(mine), 2 + max_dlas), np.nan)
    if mine:
        if make_batch is None:
            make_batch = _device_batch
        batch = make_batch(mine, device)
        min_sep = min_z_separation_kms * 1000.0 / 299792458.0
        z_dla = np.stack([make_qmc_samples(s, num_samples)[0] for s in mine])
        nhi_dla = make_qmc_samples(mine[0], num_samples)[2]
        nhi_sub = make_qmc_samples(mine[0], num_samples, sub_dla=True)[2]
        local[:, 0] = batch.nulls()
        local[:, 1] = sweep_batch(batch.eval, z_dla, nhi_sub, 1, min_sep)[0][:, 0]
        local[:, 2:] = sweep_batch(batch.eval, z_dla, nhi_dla, max_dlas, min_sep)[0]
    return gather_rows(local, total, group=group)


class _DeviceBatchAdapter:
    def __init__(self, sights, device):
        from .sightline import DeviceBatch, DeviceSightline

        self.devs = [DeviceSightline.from_state(s, device=device) for s in sights]
        self.batch = DeviceBatch(self.devs)

    def eval(self, z, nhi):
        return self.batch.eval(z, nhi)

    def nulls(self):
        return np.array([d.null() for d in self.devs])


def _device_batch(sights, device):
    return _DeviceBatchAdapter(sights, device)
