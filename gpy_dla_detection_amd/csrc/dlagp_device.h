// Device-visible descriptors shared by the kernels and the host API (libdlagp_hip.so).
//
// Vocabulary:
//   entry   one wavelength of the grid the raw absorption profile is built on
//           (padded pixels, n_u + 6, with broadening; the n_u unmasked pixels without)
//   pixel   one KEPT pixel i in [0, n): its absorption is the 7-tap (or 1-tap) window of raw
//           profile values starting at entry pos[i]
//   step    4 consecutive kept pixels = the K extent of one v_mfma_f64_16x16x4_f64
//   tile    16 columns of the shared operand P (stored as interleaved tile pairs); "w-tiles" hold vech(M_i M_i') and are weighted
//           by a^2/d, "q-tiles" hold M_i and are weighted by a r/d
//   slot    position of a pixel in the MFMA path's own pixel order: the kept pixels in order, with zero-weight
//           padding slots (y = mu = omega2 = 0, v = 1, zero operand rows: they contribute exactly nothing)
//           inserted so that every chunk of 8 slots has its convolution windows inside 64 consecutive entries,
//           whatever the pixel mask looks like (reference masks: null_gp.py:139-152, read_spec.py:38-71)
#pragma once
#include <stdint.h>

#include "../../include/dlagp_lyman.h"

#define DLAGP_TILE_COLS 16
#define DLAGP_STEP_PIXELS 4
#define DLAGP_TILE_DOUBLES (DLAGP_TILE_COLS * DLAGP_STEP_PIXELS) /* one (step, tile) block */
#define DLAGP_ROW_EXTRA 8 /* trailing scalars of a partial row: [0] sum r^2/d, [1] sum log d, [2] n log 2pi */

// One sightline, as the kernels see it.  All pointers are device pointers.
struct SightlineDev {
    int n;          // kept pixels
    int n_pad;      // n rounded up to a multiple of 8 (padding pixels contribute exactly zero)
    int nsteps;     // n_pad / 4
    int k;          // rank of M
    int num_entries;  // E
    int num_lines;  // Lyman-series members
    int ntaps;      // 7 with instrumental broadening, 1 without
    // The wavelength grid is finite, increasing and uniform in log10 to within 0.4 pixel (true for the
    // reference's grids, null_gp.py:159-177): entry e sits at log10(lambda) = grid_log0 + e / grid_inv_dlog.
    // Lets the MFMA kernel locate the few entries in a line core from the absorber redshift alone.
    int grid_uniform;
    double grid_log0, grid_inv_dlog;
    int lls_break;  // 1: add the Lyman-limit continuum opacity of every absorber (reference voigt_lls.py:258-284)
    int ntw;        // w-tiles: ceil(k (k + 1) / 2 / 16)
    int ntq;        // q-tiles: ceil(k / 16)
    int row_width;  // (ntw + ntq) * 16 + DLAGP_ROW_EXTRA doubles per partial row
    int npair;      // ceil((ntw + ntq) / 2): tile pairs per step in P
    const double *y;       // [n_pad]  padding: 0
    const double *mu;      // [n_pad]  padding: 0
    const double *omega2;  // [n_pad]  padding: 0
    const double *v;       // [n_pad]  padding: 1  (so d = 1, log d = 0, r = 0)
    const int *pos;        // [n_pad]  first entry of the pixel's window; padding: pos[n - 1]
    const double *lam;     // [E]      observed wavelength of every entry, Angstrom
    const double *P;       // [nsteps][tile pairs][64][2] shared operand of the generic path, see build_shared_operand
    // MFMA path (rank padded to 20): slot-ordered copies
    int n_slots;           // kept pixels + padding slots, a multiple of 8
    int nsteps_m;          // n_slots / 4
    const double *Pm;      // [nsteps_m][8 pairs][64][2] MFMA operand, see build_mfma_operand
    const double *aux;     // [n_slots][8] per-slot records {y, mu, omega2, v, pos, 0, 0, 0}
    double taps[DLAGP_NUM_TAPS];  // instrument profile (or {1, 0, ...})
    // Position of this sightline in the CALLER's arrays (samples in, log-likelihoods out).  Equal to its position in
    // the launch's descriptor table except when a batch that mixes line counts / broadening / Lyman-limit flags
    // runs as several homogeneous launches, each over a subset of the members (dlagp_hip.hip, LaunchGroup::parts).
    int batch_index;
};

// Where a launch finds the absorber parameters of sample (sightline b, sample s, absorber j).
//   direct:  z[(b S + s) kd + j], nhi[...]                       (dlagp_eval*: the caller laid them out)
//   indexed: absorber 0 is QMC point s, absorber j > 0 is QMC point base[b][j-1][s] drawn from the
//            previous levels' posteriors (reference dla_gp.py:135-150); z[b S + point],
//            nhi[nhi_stride b + point]                            (device-resident multi-level sweep)
struct SampleSource {
    const double *z;
    const double *nhi;
    const int *base;   // indexed mode: [nb][base_levels][S] int32
    int indexed;       // 0 = direct
    long nhi_stride;   // 0 when all sightlines share one column-density table, else S
    int base_levels;   // rows of `base` per sightline (max_dlas - 1)
    // Profile cache of a multi-level sweep (indexed mode, MFMA path).  Every absorber of every level is
    // one of the sightline's S QMC points, and sample s of level l stacks point base[l-1][s] on the
    // absorbers sample s had at level l-1 (reference dla_gp.py:135-150).  So level 1 keeps the convolved
    // absorption of every point on the sightline's pixel slots (cache_mode 1),
    // and level l > 1 forms the product  cache[s] cache[base[0][s]] .. cache[base[l-2][s]]  (cache_mode 2) instead of
    // evaluating Voigt profiles: one streamed row per sample and level, the stacked rows -- a handful of distinct ones per
    // wave -- from L2 (kernels_mfma.h).  240 MB per sightline at S = 10 000, n = 3000.
    // Layout: tiles of 16 points x 64 slots.  Point p, slot x of sightline b sits at
    //   cache[b G cache_group_stride + (p / 16) cache_group_stride + (x / 64) 1024 + (p % 16) 64 + x % 64],
    // G = ceil(S / 16) groups of points, cache_stride = 64 T = the row length padded to whole tiles,
    // cache_group_stride = 16 cache_stride: the 16 rows a wave
    // streams in one pass of a cached level are one contiguous 8 KB block, and its passes follow each other in memory.
    double *cache;
    double *running;   // unused since round 4 (no running product is kept in memory)
    long cache_stride;  // slots per row, a multiple of 64
    int cache_mode;    // 0 off, 1 store level-1 profiles, 2 multiply cached rows
    int cache_write;   // unused since round 4
    // Indexed mode on a slice of the samples (one rank's share when ONE sightline's samples are split over several
    // GPUs): the launch evaluates samples [sample_offset, sample_offset + S) of tables that hold table_S samples
    // per sightline.  table_S == 0: the launch covers the whole table (offset 0, table_S = S).
    long sample_offset;
    long table_S;
    long cache_group_stride;  // doubles from one group of 16 points to the next: 16 cache_stride
    // Pair launch (round 5).  The sub-DLA model and level 1 of the DLA model are both ONE absorber per sample at the SAME
    // redshift samples -- MATLAB: one `sample_z_dlas(i)` for both (multi_dlas/process_qsos_multiple_dlas_meanflux.m:335-373);
    // Python: dla_samples.py:94-104 and subdla_samples.py:115-125 are the same formula on the same offsets -- and differ
    // only in the column density.  One launch evaluates both: a wave's 16 MFMA rows are 8 QMC points x 2 column
    // densities, the Voigt optical depth tau(z, lambda) is formed once per point (three quarters of the sample loop) and
    // exponentiated twice.  pair_S > 0 marks such a launch: its S is 16 ceil(pair_S / 8) "virtual" samples, virtual
    // sample v = 16 block + r is point 8 block + (r & 7) of model r >> 3 (0: nhi, 1: nhi2); indexed mode, kd = 1.
    const double *nhi2;  // the second model's column densities
    long nhi2_stride;    // 0 when all sightlines share one table, else the table's stride (as nhi_stride)
    long pair_S;         // QMC points per sightline of a pair launch; 0 = an ordinary launch
};

// Many small transfers as ONE copy and one kernel.  A host <-> device transfer of a few KB is a blit kernel of its own
// (4 us on an idle device, queued behind the previous one): a context's set-up made nine of them, a one-spectrum model
// selection a dozen in and out -- 7 % of its kernel time and all of it on the caller's critical path.  The pieces are laid
// end to end in a pack buffer, which crosses the bus in one copy; `copy_segments` moves them between the pack and their
// device arrays (host-to-device: after the copy; device-to-host: in front of it).
#define DLAGP_COPY_SEGMENTS 24
struct CopySegment {
    unsigned long long device_ptr;  // the piece's device array
    unsigned long long offset;      // of the piece in the pack buffer, bytes (a multiple of 256)
    unsigned long long bytes;       // a multiple of 4
};
struct CopyTable {
    int count;
    int to_pack;  // 1: device arrays -> pack (results on their way out), 0: pack -> device arrays
    CopySegment seg[DLAGP_COPY_SEGMENTS];
};

// The level evidence folded into the finalisation (finalize_rank20's optional tail; reference dla_gp.py:155-190): every
// workgroup writes its samples' log p - log S (NaN where too_close), leaves (max, sum of exp(. - max), count) of its 16 samples
// in `part[block]`, and the LAST workgroup of a sightline to arrive (ticket) combines the blocks' triples in block order:
//   evidence = M + log(sum_b s_b exp(m_b - M) / n) - level log S,   M = max_b m_b   (NaN for an all-NaN level).
// evidence == nullptr: no tail.  tickets[nb] are zero when a launch starts and zero again when it ends.
struct EvidenceTail {
    double *evidence;                 // [nb] by batch_index
    const unsigned char *too_close;   // [nb][S] by batch_index, or nullptr
    double *part;                     // [nblocks][3] scratch
    int *tickets;                     // [nb] by the launch's sightline order
    double log_s;
    int level;
};

// What the kernels that end an occupation level need (reference dla_gp.py:155-218; kernels_common.h "level finish").
// One struct, passed by value, for the one-workgroup kernel and for the three kernels of the spread form.
//   weights  [nb][S]  exp(value - max) of the level, 0 for a NaN sample
//   exch     [nb][DLAGP_LEVEL_EXCH]  what the two phases hand each other: the 1024 virtual lanes' maxima and weight sums,
//            the 16 waves' counts of finite samples
//   ties     [nb][max_dlas - 1]  searches of the resampling that ended within DLAGP_TIE_MARGIN of a cdf step (NULL: not
//            counted): the host settles those levels with NumPy's own arithmetic (include/dlagp.h, dlagp_batch_resample_ties)
#define DLAGP_LEVEL_LANES 1024      // virtual lanes: lane t owns samples [t chunk, (t + 1) chunk), chunk = ceil(S / 1024)
#define DLAGP_LEVEL_EXCH (2 * DLAGP_LEVEL_LANES + 16)  // the lanes' maxima and sums, the 16 waves' counts of finite samples
#define DLAGP_TIE_MARGIN 1e-11      // on the cdf's scale [0, 1]: far above any rounding difference between this library's cdf
                                    // and NumPy's sequential cumsum (<= ~S eps = 1e-12 at S = 10 000), far below the spacing of
                                    // 53-bit uniforms' chance encounters (2 S margin = 2e-7 of the searches are settled on the host)
struct LevelArgs {
    const double *ll;        // [nb][S] raw log-likelihoods of the level
    long S;
    int level, max_dlas;
    SampleSource src;        // redshifts of a sample's absorbers (separation mask)
    double min_z_separation;
    double log_s;            // log(S), from the host
    const double *uniforms;  // [nb][max_dlas - 1][S]
    double *sample_ll;       // [nb][max_dlas][S]
    double *evidences;       // [nb][max_dlas]
    int *base;               // [nb][max_dlas - 1][S]
    double *weights;
    double *exch;
    int *alive;              // [nb]
    const int *status;       // the launch's non-PD flag ...
    int *sticky_status;      // ... accumulated over the sweep's levels
    int *ties;
};

// Two level finishes in one pair of launches: the rows of the grid beyond `nb_first` sightlines belong to `second` (the
// one-absorber model that rode in level 1's pair launch: evidence only, its own buffers).
struct LevelArgsTwo {
    LevelArgs first, second;
    int nb_first;
};

#if defined(__HIPCC__)
__device__ __forceinline__ long sample_point(const SampleSource &src, long b, long s, int j, long S, int kd) {
    if (src.pair_S) {  // pair launch: the point behind virtual sample s (rows beyond the last point repeat it; never stored)
        const long p = (s >> 4) * 8 + (s & 7);
        return p < src.pair_S ? p : src.pair_S - 1;
    }
    if (!src.indexed) return (b * S + s) * kd + j;
    const long T = src.table_S ? src.table_S : S;
    const long g = s + src.sample_offset;
    return j == 0 ? g : (long)src.base[(b * src.base_levels + (j - 1)) * T + g];
}
__device__ __forceinline__ double sample_z(const SampleSource &src, long b, long s, int j, long S, int kd) {
    const long p = sample_point(src, b, s, j, S, kd);
    if (src.pair_S) return src.z[b * src.pair_S + p];
    return src.indexed ? src.z[b * (src.table_S ? src.table_S : S) + p] : src.z[p];
}
__device__ __forceinline__ double sample_nhi(const SampleSource &src, long b, long s, int j, long S, int kd) {
    const long p = sample_point(src, b, s, j, S, kd);
    if (src.pair_S) return (s & 8) ? src.nhi2[b * src.nhi2_stride + p] : src.nhi[b * src.nhi_stride + p];
    return src.indexed ? src.nhi[b * src.nhi_stride + p] : src.nhi[p];
}
#endif

// Per-line constants of the Voigt profile, identical for every sightline; lives in constant memory.
//   velocity_l = lam * mult_l(z) - c,  mult_l(z) = c / (lambda_l (1 + z)) / 1e8   (voigt.py:299,305)
//   x = velocity / (sqrt2 sigma),  y_l = gamma_l / (sqrt2 sigma)                  (voigt.py:247)
//   tau_l = lc_l Re w(x + i y_l) / (sqrt(2 pi) sigma)                             (voigt.py:248,307)
struct LineConstants {
    double lambda_cm[DLAGP_MAX_LINES];   // transition wavelengths, cm
    double y[DLAGP_MAX_LINES];           // damping parameter
    double k_core[DLAGP_MAX_LINES];      // lc_l / (sqrt(2 pi) sigma)
    double k_wing[DLAGP_MAX_LINES];      // k_core * y / sqrt(pi): prefactor of the asymptotic form
    // far wing, everything folded: tau_l = u W_l(u), u = 1/x^2,
    // W_l(u) = k_wing [A(u) - y^2 u (1 + 5u + 26.25u^2)], A = near-minimax fit (faddeeva_table.h)
    double wing[DLAGP_MAX_LINES][6];
    // 2^f on [-1/2, 1/2], degree-9 fit (faddeeva_table.h).  Kept here, in constant memory, rather than as
    // literals: values that arrive through scalar loads stay in SGPRs and feed v_fma_f64 directly, whereas
    // hipcc parks literal coefficients in VGPRs (a copy per Horner step) unless forced by inline asm, and
    // every inline-asm statement costs an s_nop.
    double exp2c[10];
    double inv_sqrt2_sigma;              // 1 / (sqrt2 sigma)
    double c;                            // speed of light, cm/s
};
