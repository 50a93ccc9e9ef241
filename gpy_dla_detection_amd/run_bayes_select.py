"""
Catalogue driver: Bayesian model selection [null, sub-DLA, DLA(1..max_dlas)] for a list of spectra.

Same job and same outputs as the reference driver `run_bayes_select.process_qso`
(run_bayes_select.py:30-295): every array it writes to `processed_qsos_multi_meanflux.h5`
comes back under the same dataset name and shape.  What differs is the execution:

  * the reference loops over spectra and, per spectrum, builds three GP objects, calls `set_data` on
    each and sweeps 1 + S + max_dlas * S samples in Python (:141-245);
  * here the data half of `set_data` runs once per spectrum on the host (`null_gp.prepare_data`),
    the model half (`get_interp`) on the device (`dlagp_create_from_model`), and a batch of
    spectra is swept level by level with everything resident in HBM (`dlagp_batch_sweep`).
    The resampling consumes the global RNG exactly as the reference does after `np.random.seed(0)`
    per spectrum (:144), so `base_sample_inds` are the reference's.

Spectra arrive as arrays -- reading SDSS FITS files (`read_spec`, :147) is outside this package.
Multi-GPU: `rank` / `world_size` shard the catalogue over one process per GPU, balanced by the spectra's kept pixels
(`parallel.balanced_assignment`: a spectrum's work follows z_qso through the modelled window, null_gp.py:139-152);
every rank returns the per-quasar scalars and small vectors of the whole catalogue, in catalogue order, after ONE all-gather
(`parallel.gather_rows`); the per-sample arrays stay rank-local (`gather_samples`, `write_results`, `merge_results`).
"""
from collections import namedtuple
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
from scipy.special import logsumexp

from . import parallel
from .dla_gp import map_traceback
from .null_gp import prepare_data
from .samples import DLASamples, SubDLASamples
from .set_parameters import Parameters

LearnedModel = namedtuple(
    "LearnedModel",
    "rest_wavelengths mu M log_omega log_c_0 log_tau_0 log_beta prev_tau_0 prev_beta",
    defaults=(0.0023, 3.65),
)

# ... of which these hold a row of S samples per quasar -- 80 KB x (2 max_dlas) per quasar at S = 10 000, 99.8 % of a run's
# output -- and stay on the rank that computed them unless `gather_samples` asks otherwise (`process_qso`)
PER_SAMPLE = ("sample_log_likelihoods_dla", "base_sample_inds", "sample_log_likelihoods_lls")

# datasets of processed_qsos_multi_meanflux.h5 that hold one row per quasar (run_bayes_select.py:270-291)
PER_QUASAR = (
    "min_z_dlas", "max_z_dlas", "sample_log_likelihoods_dla", "base_sample_inds",
    "log_priors_no_dla", "log_priors_lls", "log_priors_dla",
    "log_likelihoods_no_dla", "log_likelihoods_lls", "log_likelihoods_dla",
    "log_posteriors_no_dla", "log_posteriors_lls", "log_posteriors_dla",
    "sample_log_likelihoods_lls", "MAP_z_dlas", "MAP_log_nhis", "p_dlas", "p_no_dlas", "model_posteriors",
)


def _logsumexp(a: np.ndarray) -> float:
    """
    `scipy.special.logsumexp` of a short 1-d array.  scipy's general-purpose routine costs 60 us a call (two calls per
    spectrum: a tenth of the driver's Python time); for finite input the textbook form agrees with it to the last
    bit or two, anything else (NaN of a stopped sweep, infinities) goes to scipy.
    """
    if np.isfinite(a).all():
        top = a.max()
        return top + np.log(np.exp(a - top).sum())
    return logsumexp(a)


def occupation_log_priors(prior, z_qso: float, max_dlas: int, dla_samples, subdla_samples) -> np.ndarray:
    """
    log priors [null, sub-DLA, DLA(1..max_dlas)] exactly as `BayesModelSelect.model_selection` assembles
    them (bayesian_model_selection.py:68-80 with dla_gp.py:398-426 and subdla_gp.py:311-346).
    """
    num_dlas, num_quasars = prior.less_ind(z_qso)
    ratio = num_dlas / num_quasars
    at_least = ratio ** np.arange(1, max_dlas + 1)
    at_least[:-1] = at_least[:-1] - at_least[1:]
    z_lls = getattr(subdla_samples, "_Z_lls", None)
    z_dla = getattr(subdla_samples, "_Z_dla", None)
    if z_lls is None:
        z_lls, z_dla = subdla_samples.Z_lls, subdla_samples.Z_dla
    sub = z_lls / z_dla * ratio ** np.arange(1, 2)
    log_priors = np.concatenate([[np.nan], np.log(sub), np.log(at_least)])
    log_priors[0] = np.log(1 - np.exp(_logsumexp(log_priors[1:])))
    return log_priors


def _empty_results(num: int, S: int, max_dlas: int) -> Dict[str, np.ndarray]:
    """
    The result arrays of `num` quasars.  The per-quasar values start as NaN (the reference's np.full / np.nan fills,
    run_bayes_select.py:116-139).  The three per-SAMPLE arrays -- 80 KB x (2 max_dlas) per quasar, 1 GB for 2048 quasars --
    are handed out untouched: every row is written whole when its batch finishes (`finish`), a failed batch ends the
    run, and filling them first kept the device waiting for 7 % of a 2048-spectrum run while one thread touched a
    gigabyte of fresh pages.
    """
    nan = lambda *shape: np.full(shape, np.nan)  # noqa: E731
    return dict(
        min_z_dlas=nan(num), max_z_dlas=nan(num),
        log_priors_no_dla=nan(num), log_priors_dla=nan(num, max_dlas), log_priors_lls=nan(num),
        log_likelihoods_no_dla=nan(num), log_likelihoods_dla=nan(num, max_dlas), log_likelihoods_lls=nan(num),
        log_posteriors_no_dla=nan(num), log_posteriors_dla=nan(num, max_dlas), log_posteriors_lls=nan(num),
        sample_log_likelihoods_dla=np.empty((num, S, max_dlas)),
        base_sample_inds=np.empty((num, S, max_dlas - 1), dtype=np.int32),
        sample_log_likelihoods_lls=np.empty((num, S)),
        MAP_z_dlas=nan(num, max_dlas, max_dlas), MAP_log_nhis=nan(num, max_dlas, max_dlas),
        model_posteriors=nan(num, 2 + max_dlas), p_dlas=nan(num), p_no_dlas=nan(num),
    )


class DeviceEngine:
    """
    The GPU back end of `process_qso`: learned model in HBM (`DeviceModel`), one context per spectrum
    built by `dlagp_create_from_model`, batches swept by `dlagp_batch_sweep`.  Any object with the same
    four methods can stand in (the CPU tests drive the oracle through this seam).
    """

    def __init__(self, learned: LearnedModel, params: Parameters, device: int = 0):
        from .sightline import DeviceModel

        self.params = params
        self.defer_sync = True  # contexts of a batch are waited for once, by the batch's creation (`sightline`)
        self.model = DeviceModel(*learned[:7], prev_tau_0=learned.prev_tau_0, prev_beta=learned.prev_beta,
                                 num_forest_lines=params.num_forest_lines, device=device)

    def sightline(self, state: dict, z_qso: float, broadening: bool):
        """post-`prepare_data` state of one spectrum -> resident sightline (get_interp happens here)"""
        from .sightline import DeviceSightline

        keep = ~state["pixel_mask"][state["ind_unmasked"]]  # dla_gp.py:360
        grid = state["padded_wavelengths"] if broadening else state["unmasked_wavelengths"]
        # no wait per spectrum: `batch()` below ends with the one synchronisation that covers the batch's members (each
        # wait would queue behind the waves of the sweep that is running meanwhile)
        return DeviceSightline.from_model(
            self.model, state["x"], state["y"], state["v"], state["this_wavelengths"], z_qso, grid, keep,
            num_lines=self.params.num_lines, broadening=broadening, defer_sync=self.defer_sync)

    def batch(self, sightlines):
        from .sightline import DeviceBatch

        return DeviceBatch(sightlines)

    def prepare_batch(self, table, rows, broadening: bool, offsets, sub_offsets=None):
        """
        `prepare_data` + `sightline` + `batch` + the sample redshift tables of the spectra `rows` of `table` in one native
        call (sightline.batch_prepare -> dlagp_batch_prepare); same contexts, same numbers, bit for bit.
        """
        from .sightline import batch_prepare, prepare_params

        if getattr(self, "_prepare_params", None) is None:
            self._prepare_params = prepare_params(self.params, normalize=True)
        return batch_prepare(self.model, table, rows, self._prepare_params, self.params.num_lines, broadening, offsets,
                             sub_offsets)

    def nulls(self, batch, sightlines) -> np.ndarray:
        return batch.nulls()

    def sweep(self, batch, sample_z, sample_nhi, max_dlas, min_z_separation, seed):
        return parallel.sweep_batch_device(batch, sample_z, sample_nhi, max_dlas, min_z_separation, seed)

    def model_evidences(self, batch, z_lls, nhi_sub, z_dla, nhi_dla, max_dlas, min_z_separation, seed):
        """`nulls` and the two `sweep`s in one call into the library (optional: `process_qso` falls back to the three)"""
        nb, S = z_dla.shape
        return batch.model_evidences(z_lls, nhi_sub, z_dla, nhi_dla, max_dlas, min_z_separation,
                                     parallel.resampling_uniforms(nb, max_dlas, S, seed))

    def release(self, batch, sightlines) -> None:
        batch.close()
        for s in sightlines:
            s.close()

    def close(self) -> None:
        self.model.close()


def kept_pixel_counts(params: Parameters, spectra, z_qso_list) -> np.ndarray:
    """
    n of every spectrum -- the pixels `set_data` keeps: inside the modelled rest-frame window and not masked
    (reference null_gp.py:139-152) -- without preparing the spectrum: what the sharding's cost model needs of the
    whole catalogue on every rank.
    """
    counts = np.empty(len(spectra), dtype=np.int64)
    for q, (wavelengths, _, _, pixel_mask) in enumerate(spectra):
        rest = params.emitted_wavelengths(np.asarray(wavelengths), z_qso_list[q])
        counts[q] = np.count_nonzero((rest >= params.min_lambda) & (rest <= params.max_lambda) & ~np.asarray(pixel_mask))
    return counts


def _single_piece_ramp(batch_size: int, S: int) -> tuple:
    """
    Sizes of the short batches a pipelined run starts on -- a quarter and a half of `batch_size`, or a half, or none: the
    shortest whose launches the planner keeps single-piece for DR12Q's shortest and longest sightlines (31 and 157 chunks
    of 8 pixel slots), in a one-model launch (nb ceil(S / 16) blocks) and in a pair launch (nb ceil(S / 8)).
    """
    import ctypes

    from . import _lib

    lib = _lib.load()

    def single_piece(nb: int) -> bool:
        for groups in (nb * -(-S // 16), nb * -(-S // 8)):
            for chunks in (31.0, 157.0):
                full, frac = ctypes.c_int(), ctypes.c_double()
                if lib.dlagp_plan_cut(groups, chunks, 0, ctypes.byref(full), ctypes.byref(frac)) != 0:
                    return False
                if full.value != 1 or frac.value < 1.0 - 1e-9:
                    return False
        return True

    quarter, half = batch_size // 4, batch_size // 2
    if quarter >= 1 and single_piece(quarter) and single_piece(half):
        return (quarter, half)
    if half >= 1 and single_piece(half):
        return (half,)
    return ()


def process_qso(
    spectra: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]],
    z_qso_list: Sequence[float],
    params: Parameters,
    prior,
    dla_samples,
    subdla_samples,
    learned: LearnedModel,
    max_dlas: int = 4,
    broadening: bool = True,
    min_z_separation: float = 3000.0,
    device: int = 0,
    batch_size: int = 32,
    seed: Optional[int] = 0,
    rank: int = 0,
    world_size: int = 1,
    group=None,
    qso_list: Optional[Sequence[str]] = None,
    engine=None,
    prefetch: bool = True,
    partition: str = "cost",
    native_prepare: bool = True,
    gather_samples: bool = False,
) -> Dict[str, np.ndarray]:
    """
    :param spectra: per quasar (observed wavelengths, flux, noise_variance, pixel_mask) -- what the
        reference's `read_spec(filename)` returns (run_bayes_select.py:147)
    :param learned: the learned GP model (`LearnedModel`; the reference loads it from a .mat file per spectrum)
    :param dla_samples, subdla_samples: QMC sample holders (`samples.DLASamples` / `SubDLASamples` or the *MAT loaders)
    :param prefetch: prepare the next batch in a worker thread while the current one is swept
    :param engine: back end (default `DeviceEngine(learned, params, device)`)
    :param seed: the reference re-seeds the global RNG with 0 before every spectrum (:144)
    :param gather_samples: with `world_size` > 1, all-gather the per-SAMPLE arrays (`PER_SAMPLE`) too, so that every rank
        returns every array whole (the behaviour up to round 5; 320 KB per quasar on every rank).  Default: they stay with
        the rank that computed them -- the result then carries `local_rows` (their catalogue indices), `rank` and
        `world_size`, `write_results` writes them as a per-rank shard and `merge_results` puts a catalogue back together
    :param native_prepare: prepare a batch's spectra in one call into the library (False: per spectrum in Python, the
        reference-shaped path the native one is tested against)
    :param partition: how `world_size` > 1 ranks divide the catalogue: "cost" (default) balances the modelled work
        (`kept_pixel_counts` -> `parallel.spectrum_cost` -> `parallel.balanced_assignment`, identical on every rank),
        "contiguous" gives every rank an equal count of consecutive spectra
    :return: dict keyed by the dataset names of processed_qsos_multi_meanflux.h5 (:248-295)
    """
    learned = LearnedModel(*learned)
    num = len(spectra)
    assert len(z_qso_list) == num
    S = params.num_dla_samples
    if world_size == 1:
        assignment = [np.arange(num)]
    elif partition == "cost":
        costs = parallel.spectrum_cost(kept_pixel_counts(params, spectra, z_qso_list), S, max_dlas)
        assignment = parallel.balanced_assignment(costs, world_size)
    elif partition == "contiguous":
        assignment = parallel.contiguous_assignment(num, world_size)
    else:
        raise ValueError("partition must be 'cost' or 'contiguous'")
    mine = assignment[rank]  # this rank's spectra, in catalogue order; local row i <-> spectrum mine[i]
    local = _empty_results(len(mine), S, max_dlas)
    min_sep = params.kms_to_z(min_z_separation)

    own_engine = engine is None and len(mine) > 0
    if own_engine:
        engine = DeviceEngine(learned, params, device)
    nhi_dla = np.asarray(dla_samples.nhi_samples, dtype=np.float64)
    nhi_sub = np.asarray(subdla_samples.nhi_samples, dtype=np.float64)

    # The sample redshifts of a spectrum are `z_min + (z_max - z_min) * offsets` for both sample holders
    # (samples.DLASamples._map_offsets; reference dla_samples.py:94-104, subdla_samples.py:115-125).  With the stock
    # holders the two bounds are found ONCE per spectrum and the tables of a whole batch come out of one broadcast
    # expression -- element by element the same multiply and add -- and when the two holders carry the same offsets the
    # sub-DLA table IS the DLA table (the library then shares one optical depth between the two models).  One thread
    # prepares for three that sweep: per spectrum this was 6 searches for the bounds and 2 x S-element expressions.
    stock_holders = (type(dla_samples).sample_z_dlas is DLASamples.sample_z_dlas
                     and type(subdla_samples).sample_z_lls is SubDLASamples.sample_z_lls
                     and type(dla_samples)._map_offsets is DLASamples._map_offsets
                     and type(subdla_samples)._map_offsets is DLASamples._map_offsets
                     and dla_samples.params is params and subdla_samples.params is params)
    if (type(params).min_z_dla is Parameters.min_z_dla and type(params).max_z_dla is Parameters.max_z_dla
            and type(params)._in_model_range is Parameters._in_model_range):
        z_range = params.z_dla_range  # both bounds from one in-range test
        stock_range = type(params).z_dla_range is Parameters.z_dla_range
    else:
        stock_range = False
        z_range = lambda w, z: (params.min_z_dla(w, z), params.max_z_dla(w, z))  # noqa: E731
    same_offsets = stock_holders and np.array_equal(dla_samples.offset_samples, subdla_samples.offset_samples)
    # The whole of `prepare` below -- the data half of set_data, the bounds, the tables, the contexts and the batch -- is ONE
    # call into the library when everything it restates is stock (dlagp_batch_prepare: bit-identical, and the interpreter
    # lock is free for the sweeping and finishing threads meanwhile).  Overridden `Parameters` methods, other sample holders,
    # spectra that are not float64, or an engine without the entry (the CPU tests' oracle engine) keep the Python.
    table = None
    if (native_prepare and stock_holders and hasattr(engine, "prepare_batch") and stock_range
            and all(getattr(type(params), name) is getattr(Parameters, name)
                    for name in ("emitted_wavelengths", "observed_wavelengths", "kms_to_z")) and len(mine)):
        from .sightline import SpectrumTable

        table = SpectrumTable(spectra, z_qso_list)
        if not table.ok:
            table = None

    def prepare(start: int, stop: int):
        """host index logic + device contexts of local rows [start, stop): everything a sweep needs"""
        if table is not None:
            batch, sightlines, z_dla, z_lls, scalars = engine.prepare_batch(
                table, mine[start:stop], broadening, dla_samples.offset_samples,
                None if same_offsets else subdla_samples.offset_samples)
            local["min_z_dlas"][start:stop], local["max_z_dlas"][start:stop] = scalars[:, 3], scalars[:, 4]
            return sightlines, batch, z_dla, (z_dla if same_offsets else z_lls)
        sightlines, z_dla, z_lls, bounds_z = [], [], [], []
        try:
            for i in range(start, stop):
                q = int(mine[i])
                wavelengths, flux, noise_variance, pixel_mask = spectra[q]
                z_qso = z_qso_list[q]
                rest_wavelengths = params.emitted_wavelengths(wavelengths, z_qso)
                st = prepare_data(params, rest_wavelengths, flux, noise_variance, pixel_mask, z_qso, normalize=True)
                sightlines.append(engine.sightline(st, z_qso, broadening))
                if stock_holders:
                    bounds_z.append(z_range(st["this_wavelengths"], z_qso))
                else:
                    z_dla.append(dla_samples.sample_z_dlas(st["this_wavelengths"], z_qso))
                    z_lls.append(subdla_samples.sample_z_lls(st["this_wavelengths"], z_qso))
                local["min_z_dlas"][i], local["max_z_dlas"][i] = z_range(wavelengths, z_qso)
            if stock_holders:
                z_min = np.array([b[0] for b in bounds_z], dtype=np.float64)[:, None]
                width = np.array([b[1] - b[0] for b in bounds_z], dtype=np.float64)[:, None]
                z_dla = z_min + width * dla_samples.offset_samples[None, :]
                z_lls = z_dla if same_offsets else z_min + width * subdla_samples.offset_samples[None, :]
            else:
                z_dla, z_lls = np.stack(z_dla), np.stack(z_lls)
            return sightlines, engine.batch(sightlines), z_dla, z_lls
        except BaseException:
            for s in sightlines:  # a batch that could not be built must not leave its contexts behind
                if hasattr(s, "close"):
                    s.close()
            raise

    def finish(start, stop, sightlines, batch, z_dla, nulls, ev_sub, ll_sub, ev_dla, ll_dla, base):
        """priors, posteriors, MAP values of local rows [start, stop) into the result arrays; frees their contexts"""
        rows = slice(start, stop)
        # the big per-sample arrays: one copy per batch
        local["sample_log_likelihoods_dla"][rows] = ll_dla
        local["base_sample_inds"][rows] = base.transpose(0, 2, 1)
        local["sample_log_likelihoods_lls"][rows] = ll_sub[:, :, 0]
        # MAP: the best sample of every level (np.nanargmax per spectrum and level, reference dla_gp.py:428-472),
        # found for the whole batch at once
        masked = np.where(np.isnan(ll_dla), -np.inf, ll_dla)
        best = masked.argmax(axis=1)  # (nb, max_dlas)
        if np.isneginf(masked.max(axis=1)).any():
            raise ValueError("All-NaN slice encountered")  # what np.nanargmax says about a level without a finite sample
        log_nhi = dla_samples.log_nhi_samples
        # The bookkeeping of the batch's spectra as a handful of array expressions instead of ~40 small NumPy calls per
        # spectrum (this thread holds the interpreter lock the preparing and sweeping threads also need): element for
        # element the per-spectrum expressions of the reference's loop body (tests/test_host_logic.py holds them to it).
        nb = stop - start
        log_priors = np.stack([occupation_log_priors(prior, z_qso_list[int(mine[i])], max_dlas, dla_samples, subdla_samples)
                               for i in range(start, stop)])  # the prior object is the caller's: one call per spectrum
        log_likelihoods = np.concatenate([np.asarray(nulls, dtype=np.float64).reshape(nb, 1), ev_sub, ev_dla], axis=1)
        log_posteriors = log_likelihoods + log_priors
        for name, values in (("log_priors", log_priors), ("log_likelihoods", log_likelihoods),
                             ("log_posteriors", log_posteriors)):
            local[name + "_no_dla"][rows] = values[:, 0]
            local[name + "_lls"][rows] = values[:, 1]
            local[name + "_dla"][rows] = values[:, -max_dlas:]
        every = np.arange(nb)
        for level in range(max_dlas):  # dla_gp.map_traceback with the level's best sample already known
            members = np.empty((nb, level + 1), dtype=np.intp)
            members[:, 0] = best[:, level]
            for j in range(level):
                members[:, j + 1] = base[every, j, best[:, level]]
            local["MAP_z_dlas"][rows, level, : level + 1] = z_dla[every[:, None], members]
            local["MAP_log_nhis"][rows, level, : level + 1] = log_nhi[members]
        finite = np.isfinite(log_posteriors).all(axis=1)
        top = log_posteriors.max(axis=1)
        with np.errstate(invalid="ignore", over="ignore"):
            norm = top + np.log(np.exp(log_posteriors - top[:, None]).sum(axis=1))
        for b in np.flatnonzero(~finite):  # NaN of a stopped sweep, infinities: scipy's general routine (see _logsumexp)
            norm[b] = _logsumexp(log_posteriors[b])
        posteriors = np.exp(log_posteriors - norm[:, None])
        local["model_posteriors"][rows] = posteriors
        local["p_dlas"][rows] = posteriors[:, -max_dlas:].sum(axis=1)
        local["p_no_dlas"][rows] = 1 - local["p_dlas"][rows]

    def sweeps(sightlines, batch, z_dla, z_lls):
        """the three models of one batch: null evidences, sub-DLA sweep, DLA sweep"""
        if hasattr(engine, "model_evidences"):  # one call into the library, one wait (the sub-DLA sweep draws nothing
            # from the RNG, so the uniforms are the DLA sweep's either way)
            nulls, (ev_sub, ll_sub, _), (ev_dla, ll_dla, base) = engine.model_evidences(
                batch, z_lls, nhi_sub, z_dla, nhi_dla, max_dlas, min_sep, seed)
            return nulls, ev_sub, ll_sub, ev_dla, ll_dla, base
        nulls = engine.nulls(batch, sightlines)
        # the sub-DLA model is swept before the DLA model (model list order, run_bayes_select.py:189);
        # with max_dlas = 1 it draws nothing from the RNG, so the DLA sweep sees the freshly seeded stream
        ev_sub, ll_sub, _ = engine.sweep(batch, z_lls, nhi_sub, 1, min_sep, seed)
        ev_dla, ll_dla, base = engine.sweep(batch, z_dla, nhi_dla, max_dlas, min_sep, seed)
        return nulls, ev_sub, ll_sub, ev_dla, ll_dla, base

    # A pipeline of five worker threads over the batches: one prepares (NumPy index logic, context creation on its own
    # HIP stream), THREE sweep (each call into the library runs on the calling thread's stream and releases the
    # interpreter lock: while one thread is in its transfers or back in Python, the others' kernels keep the device
    # busy -- and the small dependent kernels that end every level, finalisation and resampling, run beside another
    # batch's accumulate kernel: three threads measured 4.5 % above two, four no better), one finishes (priors,
    # posteriors, MAP, freeing contexts).  At most `in_flight` batches exist at a time.
    # `prefetch=False` -- and seed=None, whose draws come from the global RNG in catalogue order -- runs everything in
    # this thread, in order.
    bounds = [(start, min(start + batch_size, len(mine))) for start in range(0, len(mine), batch_size)]
    # A spectrum's numbers do not depend on the batch it rides in AS LONG AS the batch's launches are single-piece: one wave
    # then takes a block of 16 samples from the first pixel to the log-likelihood.  The planner cuts blocks into pieces when
    # that fills the wave slots better -- a batch of a few spectra, or of few samples -- and the sums then round differently
    # (~1e-13; base indices, exact for their likelihoods, can differ where those do).  So the pipeline's short FIRST batches
    # (the device waits while the first batch is prepared) are only as short as the planner itself keeps single-piece, asked
    # through the library (`dlagp_plan_cut`, pure host arithmetic); a short LAST batch agrees with its spectra's results in
    # any other batch to rounding, not to the bit.
    ramp = _single_piece_ramp(batch_size, S) if prefetch and seed is not None and len(mine) > 3 * batch_size >= 12 else ()
    if ramp:
        starts = [int(x) for x in np.cumsum((0,) + ramp)]
        bounds = [(starts[i], starts[i + 1]) for i in range(len(ramp))] + [
            (start, min(start + batch_size, len(mine))) for start in range(starts[-1], len(mine), batch_size)]

    def sweep_and_finish(start, stop, sightlines, batch, z_dla, z_lls):
        """whatever happens in the sweeps or the bookkeeping, the batch's contexts and workspace go back"""
        try:
            finish(start, stop, sightlines, batch, z_dla, *sweeps(sightlines, batch, z_dla, z_lls))
        finally:
            engine.release(batch, sightlines)

    if prefetch and len(bounds) > 1 and seed is not None:
        import threading

        # DLAGP_PIPELINE="prepare,sweep,finish,in_flight" overrides the thread counts (a tuning knob, not an interface)
        shape = [int(x) for x in os.environ.get("DLAGP_PIPELINE", "1,3,1,5").split(",")]
        in_flight = threading.Semaphore(shape[3])
        prepare_pool, sweep_pool, finish_pool = (ThreadPoolExecutor(max_workers=w) for w in shape[:3])
        failed = threading.Event()

        def stage_prepare(start, stop):
            if failed.is_set():
                raise RuntimeError("skipped: an earlier batch failed")
            return prepare(start, stop)

        def stage_sweep(start, stop, prepared):
            sightlines, batch, z_dla, z_lls = prepared.result()
            try:
                if failed.is_set():
                    raise RuntimeError("skipped: an earlier batch failed")
                return (start, stop, sightlines, batch, z_dla) + sweeps(sightlines, batch, z_dla, z_lls)
            except BaseException:
                failed.set()
                engine.release(batch, sightlines)
                raise

        def stage_finish(swept):
            try:
                result = swept.result()
                try:
                    finish(*result)
                finally:
                    engine.release(result[3], result[2])
            except BaseException:
                failed.set()
                raise
            finally:
                in_flight.release()

        finishing = []
        try:
            for start, stop in bounds:
                in_flight.acquire()
                if failed.is_set():  # the first failure ends the run: nothing further is started
                    in_flight.release()
                    break
                prepared = prepare_pool.submit(stage_prepare, start, stop)
                swept = sweep_pool.submit(stage_sweep, start, stop, prepared)
                finishing.append(finish_pool.submit(stage_finish, swept))
            errors = []
            for future in finishing:
                try:
                    future.result()
                except BaseException as exc:  # noqa: BLE001 -- the first one is re-raised below
                    errors.append(exc)
            real = [e for e in errors if "skipped: an earlier batch failed" not in str(e)]
            if errors:
                raise (real or errors)[0]
        finally:
            for pool in (prepare_pool, sweep_pool, finish_pool):
                pool.shutdown(wait=True, cancel_futures=True)
    else:
        for start, stop in bounds:
            sweep_and_finish(start, stop, *prepare(start, stop))
    if own_engine:
        engine.close()

    results = {}
    if world_size == 1:
        for name in PER_QUASAR:
            results[name] = local[name]  # no copy: the per-sample arrays are 80 KB x (2 max_dlas) per quasar
    else:
        # Every rank gets the per-quasar scalars and small vectors of the WHOLE catalogue -- priors, likelihoods, posteriors,
        # MAP values, p_dlas: (12 + 4 max_dlas + 2 max_dlas^2) doubles = 480 B per quasar at max_dlas = 4 -- in ONE padded
        # all-gather (RCCL over xGMI when the group is `nccl`).  The per-SAMPLE arrays are 320 KB per quasar: all-gathered they
        # would put the whole catalogue's 100 GB (300 000 quasars) on every rank's host and, padded, world_size times that
        # through device staging.  They stay rank-local (rows `local_rows`, written as per-rank shard files by
        # `write_results` and reassembled by `merge_results`; the reference's own multi-job runs are merged from per-chunk
        # files the same way, CDDF_analysis/sbatch_reunion.py:13-40) unless `gather_samples` restores the replicated form.
        small = [name for name in PER_QUASAR if name not in PER_SAMPLE]
        widths = [int(np.prod(local[name].shape[1:])) for name in small]
        packed = np.concatenate([local[name].reshape(len(mine), w) for name, w in zip(small, widths)], axis=1)
        full = parallel.gather_rows(packed, num, group=group, assignment=assignment)
        at = 0
        for name, w in zip(small, widths):
            results[name] = np.ascontiguousarray(full[:, at:at + w]).reshape((num,) + local[name].shape[1:])
            at += w
        for name in PER_SAMPLE:
            if gather_samples:
                rows = local[name].reshape(len(mine), int(np.prod(local[name].shape[1:])))
                full = parallel.gather_rows(rows, num, group=group, assignment=assignment)  # int32 rows travel as int32
                results[name] = full.reshape((num,) + local[name].shape[1:])
            else:
                results[name] = local[name]
        if not gather_samples:
            results["local_rows"] = np.asarray(mine, dtype=np.int64)  # catalogue index of every row of the per-sample arrays
            results["rank"], results["world_size"] = np.asarray(rank), np.asarray(world_size)
    # model settings and bookkeeping the reference stores next to the results (:250-268, :293-295)
    for name in ("prior_z_qso_increase", "k", "normalization_min_lambda", "normalization_max_lambda",
                 "min_z_cut", "max_z_cut", "num_dla_samples", "num_lines", "num_forest_lines"):
        results[name] = np.asarray(getattr(params, name))
    results["z_qsos"] = np.array(z_qso_list)
    if qso_list is not None:
        results["qso_list"] = np.array(list(qso_list))
    return results


def _write_one(results: Dict[str, np.ndarray], filename: str) -> str:
    try:
        import h5py
    except ImportError:
        target = filename[:-3] + ".npz" if filename.endswith(".h5") else filename + ".npz"
        np.savez_compressed(target, **results)
        return target
    with h5py.File(filename, "w") as f:
        for name, value in results.items():
            if name == "qso_list":
                f.create_dataset(name, data=np.array(value, h5py.string_dtype(encoding="utf-8")))
            else:
                f.create_dataset(name, data=value)
    return filename


def shard_filename(filename: str, rank: int) -> str:
    """`processed_qsos_multi_meanflux.h5` -> `processed_qsos_multi_meanflux.rank3.h5`"""
    stem, dot, ext = filename.rpartition(".")
    return "{}.rank{}.{}".format(stem, rank, ext) if dot else "{}.rank{}".format(filename, rank)


def write_results(results: Dict[str, np.ndarray], filename: str = "processed_qsos_multi_meanflux.h5") -> str:
    """
    Write `process_qso`'s output under the reference's dataset names (run_bayes_select.py:248-295):
    HDF5 when h5py is installed, otherwise the same names in a NumPy .npz next to it.

    A result of a multi-rank run with rank-local per-sample arrays (`local_rows` present): every rank writes its shard
    `<name>.rank<r>.<ext>` -- the per-sample arrays of its own quasars and `local_rows` -- and rank 0 also writes
    `filename` with everything else (the whole catalogue's per-quasar values).  `merge_results` reads them back.
    :return: the path this rank's main product went to (the shard, for ranks other than 0)
    """
    if "local_rows" not in results:
        return _write_one(results, filename)
    rank = int(results["rank"])
    shard = {name: results[name] for name in PER_SAMPLE + ("local_rows", "rank", "world_size")}
    written = _write_one(shard, shard_filename(filename, rank))
    if rank == 0:
        written = _write_one({name: value for name, value in results.items() if name not in PER_SAMPLE + ("local_rows", "rank")},
                             filename)
    return written


def _read_one(filename: str) -> Dict[str, np.ndarray]:
    if filename.endswith(".npz"):
        with np.load(filename, allow_pickle=False) as f:
            return {name: f[name] for name in f.files}
    import h5py

    with h5py.File(filename, "r") as f:
        return {name: f[name][()] for name in f}


def merge_results(main, shards) -> Dict[str, np.ndarray]:
    """
    One catalogue from the products of a multi-rank run: `main` = rank 0's main file (or its dict), `shards` = every
    rank's shard (paths or dicts, any order).  Returns the dict a single-process run returns -- the reference's dataset
    names, per-sample arrays whole and in catalogue order.  (Precedent: the reference merges the chunk files of its
    multi-job runs, CDDF_analysis/sbatch_reunion.py:13-40.)  The per-sample arrays are assembled here, on ONE host, when
    somebody wants them whole: 320 KB per quasar.
    """
    out = dict(main if isinstance(main, dict) else _read_one(main))
    shards = [s if isinstance(s, dict) else _read_one(s) for s in shards]
    num = out["z_qsos"].shape[0]
    covered = np.zeros(num, dtype=bool)
    for name in PER_SAMPLE:
        first = shards[0][name]
        out[name] = np.empty((num,) + first.shape[1:], dtype=first.dtype)
    for shard in shards:
        rows = np.asarray(shard["local_rows"], dtype=np.int64)
        if covered[rows].any():
            raise ValueError("two shards hold the same quasar")
        covered[rows] = True
        for name in PER_SAMPLE:
            out[name][rows] = shard[name]
    if not covered.all():
        raise ValueError("the shards cover %d of %d quasars" % (int(covered.sum()), num))
    out.pop("world_size", None)
    return out
