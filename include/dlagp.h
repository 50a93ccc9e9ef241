/*
 * dlagp.h -- C ABI of the MI355X-native GP model-evidence sweep (libdlagp_hip.so).
 *
 * The reference (jibanCat/gpy_dla_detection) has no FFI; its seam is the Python method
 * surface of NullGP / DLAGP / SubDLAGP.  Each entry point below replaces one piece of
 * that surface.  Reference paths are relative to the reference repository root.
 *
 *   dlagp_create / dlagp_destroy   the state `NullGP.set_data` + `get_interp` leave on the
 *                                  instance and that the sweep reads
 *                                  (gpy_dla_detection/null_gp.py:117-177, :240-242)
 *   dlagp_eval                     the body of the QMC loop: S x `sample_log_likelihood_k_dlas`
 *                                  (dla_gp.py:132-159 -> :311-329 -> :331-396 ->
 *                                   voigt.py:251-322 -> null_gp.py:307-360;
 *                                   subdla_gp.py:130-157 is the same code)
 *   dlagp_null                     `NullGP.log_model_evidence` (null_gp.py:294-305)
 *   dlagp_mvnpdf                   `NullGP.log_mvnpdf_low_rank` staticmethod (null_gp.py:307-360)
 *   dlagp_absorption               the absorption vector inside `this_dla_gp`
 *                                  (dla_gp.py:360-388), from which the wrapper rebuilds
 *                                  (dla_mu, dla_M, dla_omega2) (:392-394)
 *   dlagp_eval_device /
 *   dlagp_batch_*                  the same sweep on HBM-resident samples, one or many
 *                                  sightlines per launch (the reference's per-spectrum loop,
 *                                  run_bayes_select.py:141-189, made a batch)
 *   dlagp_*eval_evidence_device    ... with the occupation level's evidence from the same launch
 *                                  sequence (dla_gp.py:155-190)
 *
 * Conventions
 *   - all floating point is IEEE-754 float64; `keep` is one byte per pixel (0/1)
 *   - every function returns 0 on success, a negative DLAGP_E* code on failure; the message is
 *     available from dlagp_last_error() (thread-local).  Nothing aborts the process: every entry
 *     point is an exception barrier -- a C++ exception raised inside the library (std::bad_alloc,
 *     std::length_error, anything else) comes back as DLAGP_ENOMEM / DLAGP_EINVAL, never as an
 *     unwinding into the caller's frames (ctypes, cgo, JNI).
 *   - the caller owns all host buffers; the library copies what it needs to the device in
 *     dlagp_create and owns that device memory until dlagp_destroy.
 *   - a context is not thread-safe; use one per (sightline, host thread).
 *   - there is no CPU fallback: without a usable GPU every compute call fails with DLAGP_EDEVICE.
 */
#ifndef DLAGP_H
#define DLAGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLAGP_OK 0
#define DLAGP_EINVAL (-1)  /* bad argument (shape, NULL pointer, unsupported size)          */
#define DLAGP_EDEVICE (-2) /* no GPU, HIP runtime error, out of device memory               */
#define DLAGP_ELINALG (-3) /* B = I + M' D^-1 M not positive definite (numpy LinAlgError)   */
#define DLAGP_ENOMEM (-4)  /* out of HOST memory / a size the host cannot address (ABI >= 2)  */

#define DLAGP_MAX_ABSORBERS 8 /* absorbers per sample (the reference sweeps 1..4)            */
#define DLAGP_MAX_RANK 64     /* largest k of the low-rank factor                            */

/* bits of the `broadening` argument of dlagp_create / dlagp_create_from_model */
#define DLAGP_FLAG_BROADENING 1 /* 7-tap instrument convolution (voigt.py:311-319)          */
#define DLAGP_FLAG_LLS_BREAK 2  /* absorbers also carry the Lyman-limit continuum opacity
                                   N / 10^17.2 (lambda_rest / 911.7641 A)^3 bluewards of the
                                   limit (reference voigt_lls.py:258-284, :348; the profile of
                                   examples/gp_find_lls.py:159-224).  One absorber per sample on
                                   a log-uniform grid (and the deeper levels of a sweep, which
                                   multiply cached profiles) run the MFMA kernel, the rest the
                                   generic one */
#define DLAGP_FLAG_DEFER_SYNC 4 /* batch drivers: return without waiting for the context's uploads
                                   and set-up kernels (they run on the calling thread's set-up
                                   stream, and an event recorded behind them stays with the context).
                                   Whoever uses the context first -- dlagp_batch_create with it as a
                                   member, an evaluation, dlagp_ctx_read_model, dlagp_destroy, on ANY
                                   host thread -- waits for that event, so no rule about threads has
                                   to be kept.  Creating the contexts of a catalogue batch costs one
                                   wait instead of one per spectrum (each wait queues behind the
                                   running sweep's waves) */

typedef struct dlagp_ctx dlagp_ctx;     /* one sightline's model + data, resident in HBM    */
typedef struct dlagp_batch dlagp_batch; /* a launch group of sightlines on one device       */
typedef struct dlagp_model dlagp_model; /* the learned GP model (rest-frame grid), in HBM   */

/* ABI version of this header; bumped on any signature or status-code change.  2: DLAGP_ENOMEM.  3: near-tie counts of the
 * resampling (dlagp_batch_resample_ties; larger d_cdf_scratch / d_flags of dlagp_level_finish_device), dlagp_prepare_spectrum,
 * dlagp_batch_prepare.  (Entry points ADDED without a bump, existing ones unchanged: dlagp_eval_evidence_device,
 * dlagp_batch_eval_evidence_device.) */
int dlagp_abi_version(void);

/* Number of visible HIP devices (0 if none / runtime unusable). */
int dlagp_device_count(void);

/* Message of the last failure on the calling thread ("" if none). */
const char *dlagp_last_error(void);

/*
 * Upload one sightline.
 *   n            kept pixels (length of y, mu, omega2, v; rows of M)
 *   n_u          unmasked pixels in the modelling window, n_u >= n      (null_gp.py:139-147)
 *   k            rank of M, 1 <= k <= DLAGP_MAX_RANK                    (set_parameters.py:36)
 *   M            element (i, c) at M[i * m_row_stride + c * m_col_stride] (strides in doubles;
 *                the reference's this_M is Fortran-ordered, null_gp.py:88-93)
 *   wavelengths  observed wavelengths (Angstrom) the absorption profile is built on:
 *                n_u + 6 padded pixels if broadening != 0 (null_gp.py:159-177),
 *                else the n_u unmasked pixels                           (dla_gp.py:364-367)
 *   keep         n_u flags, keep[j] != 0 iff unmasked pixel j is a kept pixel
 *                (= ~pixel_mask[ind_unmasked], dla_gp.py:360); exactly n flags set
 *   num_lines    Lyman-series members in the Voigt profile, 1..31       (set_parameters.py:60)
 *   broadening   DLAGP_FLAG_* bits; 0 / 1 keep their meaning (no / with instrumental broadening)
 *   device       HIP device ordinal
 */
int dlagp_create(dlagp_ctx **out, int n, int n_u, int k, const double *y, const double *mu,
                 const double *M, ptrdiff_t m_row_stride, ptrdiff_t m_col_stride,
                 const double *omega2, const double *v, const double *wavelengths,
                 const uint8_t *keep, int num_lines, int broadening, int device);

void dlagp_destroy(dlagp_ctx *ctx);

/* Shape queries (for wrappers / tests). */
int dlagp_ctx_n(const dlagp_ctx *ctx);
int dlagp_ctx_k(const dlagp_ctx *ctx);
int dlagp_ctx_device(const dlagp_ctx *ctx);

/*
 * out[s] = log N(y; mu*a_s, (M*a_s)(M*a_s)' + diag(omega2*a_s^2 + v)),  s = 0..S-1, where
 * a_s = prod_j voigt_absorption(wavelengths, nhi[s*kd+j], z[s*kd+j])[keep].
 * Host pointers; synchronous.  1 <= kd <= DLAGP_MAX_ABSORBERS.
 * No "- log S" Occam term is applied (the wrapper does, dla_gp.py:157-159).
 */
int dlagp_eval(dlagp_ctx *ctx, int64_t S, int kd, const double *z, const double *nhi,
               double *out);

/*
 * The learned GP model on its rest-frame grid -- the arguments of NullGP.__init__
 * (reference null_gp.py:35-71) -- uploaded once per device and shared by every spectrum.
 *   rest_wavelengths, mu, log_omega : [n_grid]; M : [n_grid][k] with element strides as in dlagp_create
 *   log_c_0, log_tau_0, log_beta    : Lyman-series noise model (null_gp.py:216-236)
 *   prev_tau_0, prev_beta           : mean-flux suppression (null_gp.py:197-214)
 *   num_forest_lines                : series members in the effective optical depth (set_parameters.py)
 */
int dlagp_model_create(dlagp_model **out, int n_grid, int k, const double *rest_wavelengths,
                       const double *mu, const double *M, ptrdiff_t m_row_stride,
                       ptrdiff_t m_col_stride, const double *log_omega, double log_c_0,
                       double log_tau_0, double log_beta, double prev_tau_0, double prev_beta,
                       int num_forest_lines, int device);
void dlagp_model_destroy(dlagp_model *model);

/*
 * dlagp_create with NullGP.get_interp (reference null_gp.py:179-242, effective_optical_depth.py:10-80)
 * done on the device: this_mu, this_M, this_omega2 are interpolated from `model` at the rest
 * wavelengths x[n] of the kept pixels, suppressed by the Lyman-series mean flux at the observed
 * wavelengths this_wavelengths[n] and z_qso, and never visit the host.  y, v: [n] flux and noise
 * variance of the kept pixels (after NullGP.set_data's normalisation); wavelengths / keep /
 * num_lines / broadening as in dlagp_create.  A rest wavelength outside the model grid returns
 * DLAGP_EINVAL (scipy's interp1d raises ValueError there).  The context lives on the model's device.
 */
/* waits for the calling thread's pending context set-up work on `device` (optional: contexts created with
 * DLAGP_FLAG_DEFER_SYNC are waited for at their first use anyway) */
int dlagp_setup_sync(int device);

int dlagp_create_from_model(dlagp_ctx **out, const dlagp_model *model, int n, int n_u,
                            const double *x, const double *y, const double *v,
                            const double *this_wavelengths, double z_qso,
                            const double *wavelengths, const uint8_t *keep, int num_lines,
                            int broadening);

/*
 * f1, the data half.  What NullGP.set_data does with a RAW spectrum before the model is interpolated (reference
 * gpy_dla_detection/null_gp.py:117-177) -- rest wavelengths, normalisation by the median flux of the normalisation window,
 * the cut to the modelled window, masks, the grid padded by `width` log-spaced pixels per side -- and the bounds of the
 * absorber redshifts, Parameters.min_z_dla / max_z_dla (set_parameters.py:125-159).  Operation by operation the Python's
 * arithmetic: the results are bit-identical (tests/test_host_logic.py against null_gp.prepare_data), with the two library
 * calls of that arithmetic treated as follows: `median ** 2` is libm's pow as in NumPy's scalar power, and the padding pixels
 * 10 ** (log10(edge) -+ k spacing) are taken from the caller (`pad`, see dlagp_pad_callback) because NumPy dispatches log10 and
 * power to SVML on AVX-512 hosts and libm's results differ from those in the last bit for ~4 % of the arguments.
 */
typedef struct dlagp_prepare_params {
    double min_lambda, max_lambda;                               /* modelled rest-frame window   (set_parameters.py:33-34) */
    double normalization_min_lambda, normalization_max_lambda;   /* normalisation window         (:30-31)                  */
    double pixel_spacing;                                        /* log10 spacing of the padding (:58)                     */
    double lya_wavelength, lyman_limit;                          /* Angstrom                     (:16-19)                  */
    double min_z_cut, max_z_cut;                                 /* redshift differences: kms_to_z applied (:105-110)      */
    int32_t width;                                               /* padding pixels per side, 3   (:57)                     */
    int32_t normalize;                                           /* set_data(normalize=...)      (null_gp.py:121)          */
} dlagp_prepare_params;

/* pad[b][0..width) = 10 ** (log10(edges[b][0]) - (width - k) spacing) and pad[b][width..2 width) = 10 ** (log10(edges[b][1]) +
 * (k + 1) spacing), k = 0..width-1, by null_gp._logspace's arithmetic, for the nb spectra of a batch: called ONCE per
 * dlagp_batch_prepare, from the calling thread (a ctypes callback takes the interpreter lock for these few microseconds). */
typedef void (*dlagp_pad_callback)(void *user, int32_t nb, int32_t width, double pixel_spacing, const double *edges, double *pad);

/*
 * One spectrum, host only (no device is touched): the arrays `set_data` leaves on the instance.  Every output may be NULL;
 * x, y, v, this_wavelengths, keep, unmasked_wavelengths need room for num_pixels entries, padded_wavelengths for
 * num_pixels + 2 width.  pad: the 2 width padding pixels (NULL = libm's log10 / pow).
 *   scalars[7] = {normalisation median, smallest and largest unmasked wavelength, min_z_dla and max_z_dla of the kept
 *                 pixels (the sample tables'), min_z_dla and max_z_dla of the raw spectrum (the driver's min_z_dlas / max_z_dlas)}
 * DLAGP_EINVAL where the Python raises (no pixel inside the modelled window).
 */
int dlagp_prepare_spectrum(const dlagp_prepare_params *params, int num_pixels, const double *wavelengths, const double *flux,
                           const double *noise_variance, const uint8_t *pixel_mask, double z_qso, const double *pad, int *n,
                           int *n_u, double *scalars, double *x, double *y, double *v, double *this_wavelengths,
                           double *unmasked_wavelengths, double *padded_wavelengths, uint8_t *keep);

/*
 * A whole batch of raw spectra in ONE call (the caller's interpreter lock released throughout, but for `pad`): the data half
 * of every spectrum, dlagp_create_from_model for each (none waited for singly), dlagp_batch_create over them, and the
 * sample-redshift tables z_min + (z_max - z_min) * offsets (reference dla_samples.py:94-104, subdla_samples.py:115-125) --
 * what the reference driver's loop does per spectrum before the first evidence (run_bayes_select.py:141-186) and what
 * gpy_dla_detection_amd/run_bayes_select.py used to do per spectrum in Python.
 *   num_pixels[nb]; wavelengths / flux / noise_variance / pixel_mask [nb] pointers to the spectra's arrays; z_qso[nb]
 *   flags        DLAGP_FLAG_BROADENING | DLAGP_FLAG_LLS_BREAK, as dlagp_create
 *   offsets, sub_offsets [S]   QMC offsets of the DLA / sub-DLA sample holders (NULL with its table)
 *   sample_z, sub_sample_z [nb][S]  the tables (NULL = not wanted; pass sub_sample_z = NULL when both holders share offsets)
 *   scalars [nb][8]  {median, z_min, z_max, z_min of the raw spectrum, z_max of it, n, n_u, 0} (NULL = not wanted)
 * On success *out_batch and out_ctxs[0..nb) are the caller's (dlagp_batch_destroy, then dlagp_destroy each); on failure
 * nothing is left behind.
 */
int dlagp_batch_prepare(dlagp_batch **out_batch, dlagp_ctx **out_ctxs, const dlagp_model *model,
                        const dlagp_prepare_params *params, int nb, const int *num_pixels, const double *const *wavelengths,
                        const double *const *flux, const double *const *noise_variance, const uint8_t *const *pixel_mask,
                        const double *z_qso, int num_lines, int flags, dlagp_pad_callback pad, void *pad_user, int64_t S,
                        const double *offsets, const double *sub_offsets, double *sample_z, double *sub_sample_z,
                        double *scalars);

/* Copy a context's this_mu[n], this_M[n][k] (C order), this_omega2[n] back; any pointer may be NULL. */
int dlagp_ctx_read_model(dlagp_ctx *ctx, double *this_mu, double *this_M, double *this_omega2);

/* Null-model evidence: the same density with a = 1. */
int dlagp_null(dlagp_ctx *ctx, double *out);

/* Stateless log N(y; mu, M M' + diag(d)); M strided as in dlagp_create. */
int dlagp_mvnpdf(int n, int k, const double *y, const double *mu, const double *M,
                 ptrdiff_t m_row_stride, ptrdiff_t m_col_stride, const double *d, double *out,
                 int device);

/* Absorption vector a[0..n) of one sample with kd absorbers (host pointers). */
int dlagp_absorption(dlagp_ctx *ctx, int kd, const double *z, const double *nhi, double *out);

/*
 * Device-resident sweep: d_z, d_nhi ([S][kd]) and d_out ([S]) are HBM pointers on the context's
 * device; the work is enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream)
 * and NOT synchronised.  Failures of the finalisation (non-PD B) surface as NaN in d_out.
 * Stream rules of every *_device entry point: a context / batch may have launches in flight on ONE
 * stream at a time (its scratch buffers are shared by its launches; successive launches on the same
 * stream are ordered, a call that needs larger scratch drains that stream first); dlagp_destroy /
 * dlagp_batch_destroy drain the stream of the last launch before they recycle device memory.  A call on another
 * stream than the context's / batch's previous launches drains the previous stream first -- that includes the
 * host-pointer entry points (dlagp_eval, dlagp_null, dlagp_batch_eval / _null / _sweep), which run on a stream of the
 * calling thread and return with their work complete.
 */
int dlagp_eval_device(dlagp_ctx *ctx, int64_t S, int kd, const double *d_z, const double *d_nhi,
                      double *d_out, void *stream);

/*
 * Batch of sightlines evaluated by one launch sequence.  All contexts must live on the same
 * device and have the same rank.  Sample arrays are laid out [nb][S][kd], outputs [nb][S].  Members may differ in line
 * count, broadening and Lyman-limit flag: such a batch runs as one launch group per flavour on the caller's stream, results
 * at the members' positions as if it were homogeneous (dlagp_get_option "last_launch_groups").  Destroy a batch before its
 * member contexts; a context destroyed first drains the batch's launches, and the batch then refuses further calls
 * (DLAGP_EINVAL).
 */
int dlagp_batch_create(dlagp_batch **out, dlagp_ctx *const *ctxs, int nb);
void dlagp_batch_destroy(dlagp_batch *batch);
int dlagp_batch_eval_device(dlagp_batch *batch, int64_t S, int kd, const double *d_z,
                            const double *d_nhi, double *d_out, void *stream);
int dlagp_batch_eval(dlagp_batch *batch, int64_t S, int kd, const double *z, const double *nhi,
                     double *out);

/* Null-model evidence of every sightline of the batch in one launch (dlagp_null per member); out[nb]. */
int dlagp_batch_null(dlagp_batch *batch, double *out);

/*
 * The whole occupation sweep of DLAGP.log_model_evidences / SubDLAGP.log_model_evidences
 * (reference dla_gp.py:92-225, subdla_gp.py:90-223) for every sightline of the batch, with no host
 * round trip between levels: per level the accumulate + finalize launches, then one kernel that
 * subtracts log S, masks samples whose absorbers are closer than min_z_separation (:164-177), forms
 * the log-mean-exp evidence (:180-190), stops a sightline whose evidence is NaN (:200-206) and
 * resamples the base indices for the next level (:213-218).
 *
 *   sample_z   [nb][S]                 QMC redshifts of every sightline (dla_samples.sample_z_dlas)
 *   sample_nhi [S] (nhi_stride == 0, shared) or sightline b at sample_nhi + b * nhi_stride
 *   uniforms   [nb][max_dlas-1][S]     U[0,1) draws standing in for the reference's global RNG:
 *                                      np.random.random_sample(S) once per level, in level order,
 *                                      reproduces np.random.choice(S, S, p = ...) index for index
 *                                      (may be NULL when max_dlas == 1)
 *   evidences  [nb][max_dlas]          log p(D | k absorbers), NaN after an early stop
 *   sample_ll  [nb][max_dlas][S]       = sample_log_likelihoods[:, level] of the reference, transposed
 *   base_inds  [nb][max_dlas-1][S]     = base_sample_inds of the reference
 * The host variant is synchronous, sample_ll / base_inds may be NULL, and a non-PD B in any level
 * returns DLAGP_ELINALG.  The device variant takes HBM pointers (all required), enqueues on `stream`
 * and does not synchronise.  1 <= max_dlas <= DLAGP_MAX_ABSORBERS, S < 2^31.
 *
 * base_inds are np.random.choice's indices by construction, in two steps.  The library accumulates the weights in blocks
 * (1024 chunks, each relative to its own maximum, wave scans), NumPy sequentially after normalising: the two cdfs differ by rounding (~1e-16; bounded by
 * ~S eps), so an index could differ only where a uniform falls that close to a cdf step.  Every search that ends within
 * DLAGP_TIE_MARGIN = 1e-11 of the step on either side of its answer is counted per (sightline, level);
 * dlagp_batch_resample_ties returns the counts of the batch's most recent sweep, and a caller that wants the reference's
 * indices to the last bit recomputes the levels with a non-zero count with NumPy's own arithmetic (the Python classes do:
 * gpy_dla_detection_amd/sightline.py `settle_near_ties`; at most two levels in a thousand at S = 10 000 -- fewer the more the weights concentrate -- and the recomputed
 * indices have never differed outside planted tests).
 */
int dlagp_batch_sweep(dlagp_batch *batch, int64_t S, int max_dlas, const double *sample_z,
                      const double *sample_nhi, int64_t nhi_stride, double min_z_separation,
                      const double *uniforms, double *evidences, double *sample_ll,
                      int32_t *base_inds);
int dlagp_batch_sweep_device(dlagp_batch *batch, int64_t S, int max_dlas, const double *d_sample_z,
                             const double *d_sample_nhi, int64_t nhi_stride,
                             double min_z_separation, const double *d_uniforms, double *d_evidences,
                             double *d_sample_ll, int32_t *d_base_inds, void *stream);

/*
 * ties[b * (max_dlas - 1) + level] = searches of level `level`'s resampling of sightline b, in the batch's most recent
 * occupation sweep (any of dlagp_batch_sweep*, dlagp_batch_two_model_sweep_device, dlagp_batch_model_evidences*), that ended
 * within 1e-11 of a cdf step -- see above.  capacity = entries `ties` has room for (nb * (max_dlas - 1) are written; none
 * when max_dlas was 1).  After a host-pointer sweep the counts are already on the host; after a *_device sweep the call
 * waits for that sweep's stream.  DLAGP_EINVAL before the batch's first sweep.
 */
int dlagp_batch_resample_ties(dlagp_batch *batch, int32_t *ties, int64_t capacity);

/*
 * Two models at the SAME redshift samples in one device-resident call: a one-absorber model (the sub-DLA model,
 * SubDLAGP.log_model_evidences(1): sub_nhi, -> d_sub_evidences [nb], d_sub_sample_ll [nb][S]) and the occupation sweep of
 * the DLA model (as dlagp_batch_sweep_device).  In the MATLAB original both models share `sample_z_dlas(i)` and differ
 * only in the column density (multi_dlas/process_qsos_multiple_dlas_meanflux.m:335-373); the Python loaders build both
 * redshift sets by one formula (dla_samples.py:94-104, subdla_samples.py:115-125).  When the batch's shape allows (a
 * homogeneous batch on the 3-line kernel with broadening, rank <= 20) the sub-DLA sweep rides in level 1's launch: a wave's
 * 16 MFMA rows are 8 QMC points x 2 column densities and the Voigt optical depth is evaluated once per point ("pair
 * launch", option "pair_models", read-back "last_sweep_paired"); otherwise the two sweeps run one after the other.  Results
 * agree with dlagp_batch_sweep_device called twice to rounding (a sample's sums are formed the same way, but blocks of 16
 * samples are cut at other pixel positions).
 */
int dlagp_batch_two_model_sweep_device(dlagp_batch *batch, int64_t S, int max_dlas, const double *d_sample_z,
                                       const double *d_sub_nhi, int64_t sub_nhi_stride, const double *d_sample_nhi,
                                       int64_t nhi_stride, double min_z_separation, const double *d_uniforms,
                                       double *d_sub_evidences, double *d_sub_sample_ll, double *d_evidences,
                                       double *d_sample_ll, int32_t *d_base_inds, void *stream);

/*
 * The evidence loop of BayesModelSelect.model_selection (reference bayesian_model_selection.py:84-95: one
 * log_model_evidence / log_model_evidences call per model of [null, sub-DLA, DLA]) for every sightline of the
 * batch in one synchronous call: the null evidences (dlagp_batch_null), the one-absorber sweep of the sub-DLA
 * model (dlagp_batch_sweep with max_dlas = 1 over sub_sample_z / sub_sample_nhi) and the occupation sweep of the
 * DLA model (dlagp_batch_sweep) are enqueued back to back and waited for once.  Same arguments and the same
 * results as the three calls; sub_sample_ll, sample_ll, base_inds may be NULL.  (Bit for bit when the two models' launches
 * stay separate; when sub_sample_z equals sample_z by content and the batch's shape allows, the sub-DLA sweep rides in level
 * 1's launch -- see dlagp_batch_two_model_sweep_device -- and the results agree to rounding, ~1e-13 relative; base_inds can
 * then differ only where a uniform falls within that distance of a cdf step.  Option "pair_models" = 0 keeps them separate.)
 *   null_evidences [nb]   sub_evidences [nb]   sub_sample_ll [nb][S]   evidences / sample_ll / base_inds as above
 */
int dlagp_batch_model_evidences(dlagp_batch *batch, int64_t S, int max_dlas, const double *sub_sample_z,
                                const double *sub_sample_nhi, int64_t sub_nhi_stride, const double *sample_z,
                                const double *sample_nhi, int64_t nhi_stride, double min_z_separation,
                                const double *uniforms, double *null_evidences, double *sub_evidences,
                                double *sub_sample_ll, double *evidences, double *sample_ll, int32_t *base_inds);
/* The same call in two phases, for a caller that has host work of its own to do meanwhile (the drop-in classes draw the
 * occupation sweep's 30 000 uniforms from NumPy's legacy generator and assemble priors while the device already runs):
 * _start uploads the samples and enqueues the sub-DLA sweep, which needs no random numbers, and returns at once; _finish
 * takes the uniforms and the output arrays, enqueues the occupation sweep and the null evidences, waits and copies back.
 * _start followed by _finish IS dlagp_batch_model_evidences (bit for bit).  Both from the same host thread; the batch
 * must not be used in between: ANY other call on the batch between the two -- or dlagp_batch_model_evidences_cancel --
 * ends the pending call (its kernels drain), and a later _finish returns DLAGP_EINVAL. */
int dlagp_batch_model_evidences_start(dlagp_batch *batch, int64_t S, int max_dlas, const double *sub_sample_z,
                                      const double *sub_sample_nhi, int64_t sub_nhi_stride, const double *sample_z,
                                      const double *sample_nhi, int64_t nhi_stride, double min_z_separation);
int dlagp_batch_model_evidences_finish(dlagp_batch *batch, const double *uniforms, double *null_evidences,
                                       double *sub_evidences, double *sub_sample_ll, double *evidences, double *sample_ll,
                                       int32_t *base_inds);
/* gives up a _start that will not be finished (the caller's own work in between failed); a no-op without one */
int dlagp_batch_model_evidences_cancel(dlagp_batch *batch);

/*
 * Level evidence on the device (reference dla_gp.py:155-190), for sweeps whose results stay in HBM.
 * d_ll [nb][S] holds the sweep's log-likelihoods and is rewritten in place as the level's column of
 * `sample_log_likelihoods`: minus log S, NaN where d_too_close[b][i] != 0 (NULL = no mask).
 * d_evidence[b] = nanmax + log(nanmean(exp(ll - nanmax))) - level * log S.  Enqueued on `stream`.
 */
int dlagp_level_evidence_device(double *d_ll, int nb, int64_t S, int level, const uint8_t *d_too_close,
                                double *d_evidence, int device, void *stream);

/*
 * dlagp_eval_device / dlagp_batch_eval_device and dlagp_level_evidence_device as ONE call, for one occupation level whose
 * samples all sit on this device (reference dla_gp.py:132-190): d_out [nb][S] receives log p - log S (NaN where
 * d_too_close[b][i] != 0; NULL = no mask), d_evidence[b] the level's evidence.  Where the launch ends in the separate
 * finalisation kernel (one sightline or few: the partial sums of a block's pieces are added there) the evidence is computed in
 * that kernel's tail -- every workgroup leaves (max, sum of exp(. - max), count) of its 16 samples, the last one to arrive
 * combines them in block order -- and no further launch follows; otherwise (batches that finalise inside the accumulate
 * waves, ranks above 20, mixed flavours) dlagp_level_evidence_device's kernel is enqueued behind.  d_out is the same either
 * way, bit for bit; the evidence of the two forms agrees to rounding (two-stage against one-stage log-sum-exp), and each form
 * returns the same bits run after run.  dlagp_get_option("last_evidence_folded") tells which one ran; option "fold_evidence"
 * = 0 forces the second.
 */
int dlagp_eval_evidence_device(dlagp_ctx *ctx, int64_t S, int kd, const double *d_z, const double *d_nhi, double *d_out,
                               int level, const uint8_t *d_too_close, double *d_evidence, void *stream);
int dlagp_batch_eval_evidence_device(dlagp_batch *batch, int64_t S, int kd, const double *d_z, const double *d_nhi,
                                     double *d_out, int level, const uint8_t *d_too_close, double *d_evidence, void *stream);

/*
 * One sightline's samples split over several GPUs (by-sample partition).  The samples of an
 * occupation level are independent (reference dla_gp.py:132-159; MATLAB parfor,
 * multi_dlas/process_qsos_multiple_dlas_meanflux.m:333) but the resampling needs all of them (:209-218).  Per level:
 *   dlagp_batch_level_eval_device  every rank evaluates samples [s_begin, s_begin + s_count) of level `level`
 *                                  (absorber 0 = QMC point s, absorber j = point base_inds[b][j-1][s], tables of
 *                                  S_total samples per sightline) -> d_out[nb][s_count], raw log-likelihoods;
 *   (caller) all-gather the slices into d_ll[nb][S_total] -- RCCL, S_total / G doubles per rank;
 *   dlagp_level_finish_device      every rank: minus log S, separation mask, log-mean-exp evidence, resampling of
 *                                  base_inds[b][level][:] with the SAME uniforms (:155-218) -- identical indices on
 *                                  every rank, no broadcast.
 * dlagp_sweep_state_init_device NaN-/zero-fills sample_ll [nb][max_dlas][S], evidences [nb][max_dlas], base_inds
 * [nb][max_dlas-1][S] and the int32 flags before level 0.
 * d_flags: nb + 2 + nb * (max_dlas - 1) int32 -- alive per sightline, sticky non-PD status, a zero, and (ABI 3) the
 * near-tie counts [nb][max_dlas-1] of the resampling (see dlagp_batch_sweep: every rank holds the same counts and settles
 * the same levels).  d_cdf_scratch: nb * (S + DLAGP_LEVEL_EXCHANGE) doubles (ABI 3: the weights and what the kernels of a
 * level hand each other).  All pointers are HBM addresses; everything is enqueued on `stream`.
 */
#define DLAGP_LEVEL_EXCHANGE 2064
int dlagp_batch_level_eval_device(dlagp_batch *batch, int64_t S_total, int64_t s_begin, int64_t s_count, int level,
                                  const double *d_sample_z, const double *d_sample_nhi, int64_t nhi_stride,
                                  const int32_t *d_base_inds, int base_levels, double *d_out, void *stream);
int dlagp_sweep_state_init_device(int nb, int64_t S, int max_dlas, double *d_sample_ll, double *d_evidences,
                                  int32_t *d_base_inds, int32_t *d_flags, int device, void *stream);
int dlagp_level_finish_device(int nb, int64_t S, int level, int max_dlas, const double *d_ll, const double *d_sample_z,
                              double min_z_separation, const double *d_uniforms, double *d_sample_ll,
                              double *d_evidences, int32_t *d_base_inds, double *d_cdf_scratch, int32_t *d_flags,
                              int device, void *stream);

/*
 * Tuning / measurement hooks.
 *   dlagp_set_option: "pixel_splits" (0 = automatic work decomposition, h > 0 = every block of 16 samples
 *                     is cut into h equal pixel ranges), "path" (0 = automatic, 1 = generic
 *                     kernel, 2 = MFMA kernel), "timing" (1 = attach a pair of HIP events to the
 *                     dispatch of the accumulate kernel of every launch), "debug" (bit 1: launch the
 *                     stamped instantiation of the MFMA kernel, see dlagp_debug_read),
 *                     "core_patch" (1 = the MFMA kernel evaluates the line cores outside its sample
 *                     loop when it can: 3 lines, 1 absorber per sample, log-uniform wavelength grid;
 *                     0 = always test for them inside the loop),
 *                     "pool_mb" (device buffers of destroyed contexts are parked, up to this many MB,
 *                     and reused by later contexts of the same size instead of hipFree / hipMalloc;
 *                     default 2048, 0 frees everything parked and turns the pool off),
 *                     "profile_cache" (1 = dlagp_batch_sweep keeps the level-1 absorption profiles
 *                     of all QMC points in HBM, 8 S n_pad bytes per sightline, and multiplies cached
 *                     rows at the deeper levels instead of re-evaluating Voigt profiles; 0 = recompute),
 *                     "profile_cache_max_mb" (upper bound of that allocation, default 196608; never more than half of the free HBM)
 *                     "plan_item_cost" (fixed cost of a work item in the planner's model, tenths of a chunk, default
 *                     20), "plan_slots" (wave slots the planner fills, default 2048 = 256 CUs x 4 SIMDs x 2 waves),
 *                     "plan_rest_first" (dispatch remainder pieces first; tuning experiments only),
 *                     "finalize_row16_blocks" (launches of up to this many blocks of 16 samples factor with 16 lanes
 *                     per sample, larger ones with one lane per sample: identical results; default 2^40 = all, 0 = never),
 *                     "spare_workspaces" (sweep workspaces of destroyed batches kept per device for the next
 *                     batches, default 5), "plan_xcd_affine" (1 = batches of >= 8 sightlines deal their work items so
 *                     that the workgroups of one XCD sweep one sightline's operand at a time -- only where the mean sightline has at least
 *                     "plan_affine_min_chunks" chunks of 8 slots (default 256: an operand of 4 MB, one XCD's L2): on shorter
 *                     sightlines the plain (sightline, block) order, longest first, is as fast at equal lengths and faster on
 *                     ragged batches -- catalogues -- where the affine rounds leave the XCDs with unequal work; default 1),
 *                     "shared_work_stream" (1 = the kernels of the host-pointer entry points of ALL host threads go to
 *                     one stream per device, each call waiting for its own completion event: launches of different threads
 *                     run one after the other, transfers and host work overlap them; default 1 since round 5; 0: every
 *                     calling thread launches on its own stream), "shared_rows" (1 = on the cached levels of a sweep a wave whose
 *                     16 samples stack the same cached profile reads it once per pass; 0 = always 16 reads; identical
 *                     results; default 1), "fuse_finalize" (1 = when the launch plan gives every block of 16 samples ONE
 *                     piece -- batches: more blocks than wave slots -- the accumulate wave also factors its block and
 *                     writes log p itself, and no partial sums cross HBM; 0 = always the separate finalisation kernel;
 *                     identical results, bit for bit; default 1), "pair_models" (1 = dlagp_batch_model_evidences* and
 *                     dlagp_batch_two_model_sweep_device evaluate the sub-DLA model and level 1 of the DLA model in ONE
 *                     launch when both sample the same redshifts -- one Voigt optical depth per QMC point; 0 = two
 *                     launches; identical results at equal cuts, to rounding otherwise; default 1),
 *                     "level_spread" (1 = the kernels that end an occupation level -- Occam term and separation mask,
 *                     evidence, resampling -- run as two launches that spread a sightline over ~30 compute units; 0 = one
 *                     1024-thread workgroup per sightline; identical results, bit for bit; default 1), "level_search_groups"
 *                     (workgroups per sightline that share the searches of the spread form, 1..64, default 16),
 *                     "pack_transfers" (1 = the small host <-> device transfers of a host-pointer call, up to 8 MB in all,
 *                     cross the bus as ONE copy each way and a kernel distributes / collects the pieces -- a dozen dependent
 *                     4 us copies per one-spectrum model selection otherwise; 0 = one copy per array; default 1),
 *                     "null_side" (1 = in dlagp_batch_model_evidences* on a batch of up to four sightlines the null model's
 *                     two kernels run beside the sweeps, on the calling thread's transfer stream with buffers of their own,
 *                     instead of as the call's last launch: -4.5 % wall per one-spectrum model selection; identical results;
 *                     default 1), "null_side_level" (... behind an event that fires when this occupation level's evaluations
 *                     are done, so that they run beside that level's small finishing kernels and take no compute unit from an
 *                     accumulate launch: 0-based level, negative = counted from the deepest, -1 = the deepest; default -2),
 *                     "pack_setup_uploads" (the same for the arrays of a context's set-up: 0 = one copy per array, 1 = together,
 *                     2 = together when the device's kernel stream is idle at that moment -- beside a running sweep the single
 *                     copies get through sooner than a kernel finds a wave slot; default 2)
 *   dlagp_timing_read: sum of accumulate-kernel durations (ms) and launch count since the last
 *                     reset; synchronises the recorded events.
 */
int dlagp_set_option(const char *name, int64_t value);
/* Reads an option back, or one of the read-only diagnostics of the most recent launch on this process:
 * "last_launch_path" (1 = generic kernel, 2 = MFMA kernel), "last_launch_items" (waves of the last MFMA
 * launch), "last_launch_parts" (most pieces a block of 16 samples was cut into), "last_sweep_cached"
 * (1 = the last dlagp_batch_sweep* used the profile cache), "last_launch_patch" (how the last MFMA launch handled
 * the line cores: 0 tested inside the sample loop, 1 repaired once per wave, 2 repaired pass by pass),
 * "last_launch_groups" (homogeneous launch groups the last call ran as: > 1 for a batch that mixes line counts,
 * broadening or Lyman-limit flags), "last_launch_fused" (1 = the accumulate waves of the last launch finished their own
 * blocks, see "fuse_finalize"), "big_allocs" / "big_frees" (sweep-workspace buffers above the pool's 64 MB
 * limit allocated / freed since the library was loaded -- a free synchronises the device; a pipelined run in its steady
 * state shows neither), "work_stream_idle" (1 = the current device's shared kernel stream has nothing queued or running
 * at this instant: polled from a side thread it says how often the host leaves the device without work), "last_level_spread"
 * (1 = the most recent level finish ran in its spread form, see "level_spread"). */
int dlagp_get_option(const char *name, int64_t *value);
/* The work decomposition the MFMA path would choose for `groups` blocks of 16 samples of sightlines with `chunks`
 * chunks of 8 pixel slots each (pure host arithmetic, no device needed): every block is cut into `full` pieces of
 * `frac` of its chunks each, plus one remainder piece when full * frac < 1.  forced_splits as option "pixel_splits". */
int dlagp_plan_cut(int64_t groups, double chunks, int64_t forced_splits, int *full, double *frac);
/* The launch planner of the MFMA path on a batch of pixel masks, pure host code (no device needed; the same function
 * is exported by the host-only sanitizer build libdlagp_plan_host.so, csrc/plan_host_debug.cpp).  keep[b] is the mask
 * of sightline b over its n_u[b] unmasked pixels (= ~pixel_mask[ind_unmasked], reference null_gp.py:139-152), ntaps 7
 * with instrumental broadening and 1 without, S the samples per sightline, options NULL (the library's current values
 * are NOT used: defaults) or {pixel_splits, plan_item_cost, plan_rest_first, plan_slots, plan_xcd_affine, column groups
 * (1 for rank <= 20; ranks 21..64 run every block once per group of 210 + 20 columns), plan_taper (pieces per block for the
 * sightlines of a batch's last round of eight; 0 / 1 = off)}.
 *   summary[8]  {slots of all sightlines, passes, work items, blocks, most pieces per block, full pieces, remainder
 *               piece 0/1, fraction of a sightline per full piece x 1e9}
 *   slot_pixel / slot_pos [slots]   per sightline, concatenated: kept pixel of every slot (-1 = padding slot) and
 *               first entry of its convolution window
 *   passes [passes][3]  {entry_start, chunk_begin, nchunks};  items [items][6]  {sightline, block, piece, pass_begin,
 *               pass_end, column group} in dispatch order;  counts [blocks]  pieces of block b * ceil(S / 16) + group
 * Every output array may be NULL or too small (its *_cap says how many records fit): the summary then tells the sizes. */
int dlagp_plan_debug(int nb, const int *n_u, const uint8_t *const *keep, int ntaps, int64_t S, const int64_t *options,
                     int64_t *summary, int32_t *slot_pixel, int32_t *slot_pos, int64_t slots_cap, int32_t *passes,
                     int64_t passes_cap, int32_t *items, int64_t items_cap, uint8_t *counts, int64_t counts_cap);
int dlagp_timing_read(double *total_ms, int64_t *launches, int reset);
/* Diagnostic build aid: with option "debug" bit 1 set the MFMA kernel sums shader-clock stamps per
 * phase over all waves (current device): [0] Voigt loop, [1] convolution, [2] weights + MFMA loop,
 * [3] epilogue, [4] pass loop + epilogue, [5] waves, [6] prologue.  Not for production launches. */
int dlagp_debug_read(uint64_t *out, int n, int reset);
/* Diagnostic: the Voigt profile V(v; sigma, gamma_line) of reference voigt.py:241-248 (replaces scipy.special.wofz
 * there) at n velocities [cm/s] (host pointers), evaluated exactly as the kernels evaluate it: form 0 = generic
 * kernels (asymptotic series / Taylor table), form 1 = MFMA kernel (per-line wing fit / Taylor table).  Lets the
 * tests pin the device Faddeeva evaluation to the reference's golden grid directly. */
int dlagp_debug_voigt(int form, int line, int64_t n, const double *velocity, double *out, int device);

#ifdef __cplusplus
}
#endif
#endif /* DLAGP_H */
