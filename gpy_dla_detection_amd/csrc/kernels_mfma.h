// MFMA evaluation path for rank k <= 20 (zero-padded to 20): the production kernel.
//
// On gfx950 the FP64 matrix and FP64 vector instructions share one execution pipe
// (profiles/r01_microbench_dp_pipe.txt: an MFMA wave and an FMA wave on one SIMD take the SUM
// of their times), so there is no "free" VALU time under the MFMAs.  The kernel is therefore
// built to (a) spend as few VALU instructions per (sample, pixel) as possible and (b) never
// stall: every wave is independent -- no workgroup barriers, no shared staging -- and streams
// its operands from L2 one step ahead of use.
//
// Work decomposition (DESIGN.md "accumulate_mfma", "Work decomposition"):
//   one wave (= one 64-thread workgroup) is one work item of the launch plan: 16 QMC samples of one
//   sightline over a range of its chunks (dlagp_hip.hip build_plan).
//   The Gram/projection contraction is the FP64 GEMM  [samples x pixels] . [pixels x 230 cols]:
//     A operand (16 samples x 4 pixels per MFMA) = the per-pixel weights w = a^2/d (vech(M_i M_i'),
//       210 columns) and q = a r/d (M_i, 20 columns), generated in registers by the same lane that
//       feeds them to the MFMA -- they never touch memory;
//     per step 14 v_mfma_f64_16x16x4_f64 (13 w-tiles: vech columns 0..207, one q-tile: M columns 0..15)
//       and two v_mfma_f64_4x4x4_4b_f64 (vech columns 208, 209 with w; M columns 16..19 with q): the
//       4-block form takes 16 cycles, its A-operand lane layout is the 16x16x4 form's, and it spares
//       the two mostly empty 64-cycle tiles (profiles/r02_microbench_mfma_f64_4x4.txt);
//     B operand = the sightline's MFMA operand Pm (kernels_common.h build_mfma_operand), read straight
//       from L2 into registers as 8 double2 per lane and step; each register pair is reloaded for the
//       next step right after its two MFMAs have issued (a rolling one-step-ahead prefetch).
//   Accumulators: 14 tiles x 4 doubles + 2 doubles per lane (116 VGPRs) -> 2 waves per SIMD.
//
// Per pass (<= 56 kept pixels whose 7-tap windows fit in 64 consecutive entries):
//   phase 1  lane = entry, loop over the wave's 16 samples: optical depth of every absorber and
//            line (wing polynomial; Taylor/table core path behind a wave-uniform branch),
//            raw = exp(-N tau) -> LDS raw[sample][entry]            (voigt.py:299-309)
//   phase 2  lane = (sample = lane & 15, pixel slot = lane >> 4), per step of 4 pixels:
//            a = sum_t taps_t raw[pos + t] (voigt.py:319, dla_gp.py:388), d = omega2 a^2 + v,
//            r = y - mu a, w = a^2/d, q = a r/d, running sum r^2/d and product of d
//            (dla_gp.py:392-394, null_gp.py:329-333,356), then the 16 MFMAs   (null_gp.py:339-354)
// Output: partial sums part[piece][block][native layout, kernels_common.h] for finalize_rank20<.., NATIVE = true>.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "dlagp_device.h"
#include "fastmath_device.h"
#include "kernels_common.h"
#include "plan_host.h"
#include "voigt_device.h"

namespace dlagp {

typedef double d4 __attribute__((ext_vector_type(4)));

#define DLAGP_MFMA_NTW DLAGP_MFMA_W_TILES    // 13 full tiles weighted by w
#define DLAGP_MFMA_NT DLAGP_MFMA_BIG_TILES   // 14 full tiles; two 4x4x4 MFMAs carry the remaining 2 + 4 columns
#define DLAGP_RAW_STRIDE 67      // doubles per sample row in LDS; odd -> conflict-free ds_read2_b64 windows
#define DLAGP_BUFFER_FLAGS 0x00020000  // word 3 of a raw buffer resource on gfx9-family targets
// DLAGP_MFMA_GROUP (16 samples per wave), DLAGP_PASS_ENTRIES, DLAGP_CHUNK_PIXELS, DLAGP_MAX_PASS_STEPS / _CHUNKS and the
// PassDesc / WorkItem records of the launch plan: plan_host.h (shared with the host-only planner build)

// Per-(sample, absorber) constants, computed by each wave for its 16 samples into LDS:
//   m'_l = c / (lambda_l (1 + z)) / 1e8 / (sqrt2 sigma)   so that  x_l = lam m'_l - c / (sqrt2 sigma)
// (voigt.py:299,305,247 folded into one FMA per line), followed by N log2(e) (the profile is then
// 2^(-N log2e tau), voigt.py:309).
// Layout: prep_lds[sample][absorber][num_lines + 1].

// Far-wing term of one line, valid for x^2 >= 256; for smaller x^2 (and NaN) the reciprocal is clamped
// so that the value stays finite and reproducible -- the caller replaces it on the core path.
__device__ __forceinline__ double wing_term(double x2, const double *__restrict__ w) {
    const double u = fmin(fast_rcp1(x2), 1.0 / (DLAGP_FAD_XSPLIT * DLAGP_FAD_XSPLIT));
    double poly = w[DLAGP_WING_DEGREE];
#pragma unroll
    for (int m = DLAGP_WING_DEGREE - 1; m >= 0; --m) poly = fma(poly, u, w[m]);
    return u * poly;
}

// Optical depth per unit column density at one entry, all lines (voigt.py:304-309 without the
// sign and the N factor).  `mult[l]` are wave-uniform.  Every line first gets its far-wing term
// unconditionally; if any lane of the wave sits within 16 Doppler widths of a line centre
// (~6 entries per line and absorber) a wave-uniform branch swaps that lane's term for the
// Taylor/table evaluation.
// `w3v`: the wing polynomials' degree-3 coefficients of the first three lines in vector registers (LINES == 3
// only): with one operand of the first Horner step in a VGPR that step is a single v_fma_f64.
// CORE = false (3 lines only): no core test and no clamping at all -- a lane inside a line core produces
// garbage (its shared reciprocal overflows) and the caller overwrites that entry afterwards.
template <int LINES, bool CORE = true>
__device__ __forceinline__ double optical_depth(double lam, const double *__restrict__ mult, int num_lines,
                                                const double *w3v = nullptr, const double *xs_in = nullptr) {
    const int L = LINES > 0 ? LINES : num_lines;
    const double x_offset = -DLAGP_LINES.c * DLAGP_LINES.inv_sqrt2_sigma;
    constexpr double split2 = DLAGP_FAD_XSPLIT * DLAGP_FAD_XSPLIT;
    double tau = 0.0;
    if (LINES == 3) {
        // one reciprocal for the three lines: 1/X_l = (1/(X_0 X_1 X_2)) * (product of the other two),
        // with every X clamped to >= 256 so that a lane inside one line's core still gets the other
        // two lines right (its own clamped term is replaced below)
        double xs[3], xc[3], us[3];
        bool any_core = false;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            xs[l] = xs_in ? xs_in[l] : fma(lam, mult[l], x_offset);
            const double x2 = xs[l] * xs[l];
            if (CORE) any_core |= !(x2 >= split2);  // also catches NaN
            xc[l] = CORE ? fmax(x2, split2) : x2;
        }
        const double p01 = xc[0] * xc[1];
        const double r = fast_rcp1(p01 * xc[2]);
        const double r2 = r * xc[2];
        us[0] = r2 * xc[1];
        us[1] = r2 * xc[0];
        us[2] = r * p01;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const double *w = DLAGP_LINES.wing[l];
            static_assert(DLAGP_WING_DEGREE == 4, "first Horner step below pairs the coefficients 4 and 3");
            double poly;
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(poly) : "s"(w[4]), "v"(us[l]), "v"(w3v[l]));
#pragma unroll
            for (int m = DLAGP_WING_DEGREE - 2; m >= 0; --m) poly = fma(poly, us[l], w[m]);
            tau = fma(us[l], poly, tau);
        }
        if (CORE && __builtin_amdgcn_ballot_w64(any_core)) {
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const double x = xs[l];
                if (!(x * x >= split2)) {
                    const double *w = DLAGP_LINES.wing[l];
                    double poly = w[DLAGP_WING_DEGREE];
#pragma unroll
                    for (int m = DLAGP_WING_DEGREE - 1; m >= 0; --m) poly = fma(poly, us[l], w[m]);
                    tau = fma(-us[l], poly, tau);  // take the clamped wing term back out
                    const double term = DLAGP_LINES.k_core[l] * core_taylor(x, DLAGP_LINES.y[l]);
                    if (term == term) tau += term;  // np.nansum (voigt.py:309)
                }
            }
        }
    } else if (LINES > 0) {
        double xs[LINES > 0 ? LINES : 1];
        bool any_core = false;
#pragma unroll
        for (int l = 0; l < LINES; ++l) {
            const double x = fma(lam, mult[l], x_offset);
            xs[l] = x;
            const double x2 = x * x;
            any_core |= !(x2 >= split2);  // also catches NaN
            tau += wing_term(x2, DLAGP_LINES.wing[l]);
        }
        if (__builtin_amdgcn_ballot_w64(any_core)) {
#pragma unroll
            for (int l = 0; l < LINES; ++l) {
                const double x = xs[l];
                const double x2 = x * x;
                if (!(x2 >= split2)) {
                    tau -= wing_term(x2, DLAGP_LINES.wing[l]);
                    const double term = DLAGP_LINES.k_core[l] * core_taylor(x, DLAGP_LINES.y[l]);
                    if (term == term) tau += term;  // np.nansum (voigt.py:309)
                }
            }
        }
    } else {
        for (int l = 0; l < L; ++l) {
            const double x = fma(lam, mult[l], x_offset);
            const double x2 = x * x;
            double term = wing_term(x2, DLAGP_LINES.wing[l]);
            if (!(x2 >= split2)) term = DLAGP_LINES.k_core[l] * core_taylor(x, DLAGP_LINES.y[l]);
            if (term == term) tau += term;
        }
    }
    return tau;
}

// Pointers read out of the descriptor table are generic to the compiler, which then emits flat_*
// loads whose completion can only be awaited with vmcnt(0) + lgkmcnt(0) -- that serialises every
// prefetch.  These typedefs pin them to the global address space.
typedef const __attribute__((address_space(1))) double *gptr_d;
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v2d *gptr_d2;
typedef const __attribute__((address_space(1))) v4d *gptr_d4;

// ... and to scalar registers: the value is the same in every lane, but having come out of a
// vector load the compiler would otherwise keep it (and every address derived from it) in VGPRs.
template <typename G, typename T>
__device__ __forceinline__ G as_global(const T *p) {
    const unsigned long long bits = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bits);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(bits >> 32));
    return (G)(__attribute__((address_space(1))) const void *)(((unsigned long long)hi << 32) | lo);
}

// Diagnostic cycle stamps (debug bit 1): per-phase shader-clock totals over all waves, read back with
// dlagp_debug_read.  Slots: 0 Voigt sample loop, 1 convolution, 2 weights + MFMA loop, 3 epilogue,
// 4 pass loop + epilogue, 5 wave count, 6 prologue.  Compiled only into the STAMPS instantiations of the kernel.
__device__ unsigned long long DLAGP_STAMPS[8];
#ifdef DLAGP_TRACE_BUILD
// diagnostic build only (tools/wave_timeline.py): per work item {start, end (100 MHz real time), HW_ID | XCC_ID << 32, passes}
__device__ unsigned long long DLAGP_TRACE[4 * 8192];
#endif

__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// Orders this wave's LDS traffic: the LDS executes one wave's operations in issue order, so all
// that is needed between a write and another lane's read is (a) that the compiler keeps the
// program order and (b) an lgkmcnt wait.  A workgroup-scope fence would also drain vmcnt, i.e.
// wait for the global prefetches in flight.
__device__ __forceinline__ void lds_wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// grid: one wave (= workgroup) per work item of the launch plan.
// BROADEN: 7-tap instrument convolution (true) or the raw profile itself (false).
// Any number of absorbers kd >= 1: each absorber's profile is convolved and multiplied into the LDS
// buffer a[sample][pixel of the pass] before the MFMA loop (dla_gp.py:370-388).
// Dynamic LDS: the wave's per-sample constants, 16 samples x kd x (L + 1) doubles.
// MODE (profile cache of the multi-level sweep, SampleSource::cache): 0 = none; 1 = level 1, also stores the
// convolved profiles; 2 = deeper levels, phase 1 multiplies cached rows instead of evaluating Voigt profiles
// (LINES / BROADEN are then irrelevant: instantiated once, as <3, true, false, 2>).
//
// PATCH (LINES == 3, SightlineDev::grid_uniform; 1: one absorber per sample, 2: several): the line cores are
// taken out of the sample loop.
// Within 16 Doppler widths of a line centre -- at most 6 consecutive entries per (sample, line) -- the far-
// wing form does not hold and the Taylor/table evaluation (core_taylor) is needed.  Testing for that in
// the loop costs three compares per sample and entry plus a wave-wide branch that is taken for ~7 % of the
// (sample, pass) pairs with 1-6 active lanes -- 6 % of the kernel (ab/nocore experiment).
// Instead the loop runs the bare wing form everywhere (no test, no clamp: core entries come out as garbage),
// and the wave replaces the few core entries afterwards: once per wave, for every (sample, line) whose
// centre lies in the wave's pixel range, the 7 candidate entries around the centre (located through the
// log-uniform grid) are examined 64 at a time and those inside the core get their exact profile value
// 2^(-N log2e tau) (core form for that line, wing form for the other two); every pass copies the values
// that fall into its window over raw[].  With several absorbers per sample (PATCH == 2) a table of that size per
// absorber does not fit the wave's LDS: only the first candidate entry of every (sample, absorber, line) is kept,
// and each pass repairs the cores of the absorber it has just evaluated on the spot.
//
// LLS (3-line kernels): every absorber also carries the Lyman-limit continuum opacity of reference
// voigt_lls.py:258-284, N / 10^17.2 (lambda_rest / 911.7641)^3 where lambda_rest = fl(lambda / (1 + z)) is not
// above the limit.  Per sample two more constants: the scaled reciprocal that turns lambda into
// (lambda_rest / limit) 10^(-17.2/3), and the smallest grid wavelength T with fl(T / (1 + z)) > limit, found
// by stepping ulps around limit (1 + z), so that the in-loop test `lambda >= T` reproduces the reference's
// `rest_wavelengths > lambda_Lyman_limit` (:280) exactly without a division per entry.
//
// FUSED (launch plans with ONE piece per block: batches of sightlines): the wave also finishes its block -- the
// factorisation, solve and log-density of finalize_rank20 -- and writes log p to `out`; its partial sums never leave the
// chip.  A block's sums are 33 KB: written and read back by a second kernel they are 676 MB per level of a batch of 32
// (2.9 TB/s, HBM-bound, 235 us) -- 2.6 % of a level at n = 3000 pixels but 10 % at the 685 of a median DR12Q sightline and
// 19 % at 300 (profiles/r05_frac_vs_n.json).  See the epilogue.
// PAIR: a pair launch (SampleSource::pair_S, dlagp_device.h) -- the wave's rows r and r + 8 are the SAME
// QMC point under two column densities (sub-DLA model / DLA model): the sample loop runs 8 times, forms the optical depth
// once and exponentiates it twice; level 1's cache store keeps the DLA model's rows only; the finalisation -- the wave's
// own (FUSED) or finalize_rank20 -- writes rows 0..7 to the first model's output, rows 8..15 to the second's.
template <int LINES, bool BROADEN, bool STAMPS, int MODE, int PATCH, bool LLS, bool FUSED, bool PAIR>
__device__ __forceinline__ void accumulate_mfma_body(
    const SightlineDev *__restrict__ sightlines, const PassDesc *__restrict__ passes,
    const WorkItem *__restrict__ items, int nb, long S, int kd,
    SampleSource src, double *__restrict__ part, int *__restrict__ status, int launch_flags, double *__restrict__ out,
    double *__restrict__ out2) {
    constexpr bool CACHED = MODE == 2;
    constexpr int NPOOL = 16;  // distinct cached rows a wave holds per pass (cached levels)
    // Pixel slots a pass holds.  Voigt passes: 7 chunks, what a 64-entry profile window covers behind the 7-tap
    // convolution.  The cached levels have no window: lane = pixel slot, so a pass is 8 chunks = all 64 lanes, and the
    // kernel walks its work item's chunk range in passes of its own (same chunks, same steps in the same order -- the
    // sums stay bit-identical to the recomputing kernel's -- in 1/8 fewer passes).
    constexpr int PASS_CHUNKS = CACHED ? DLAGP_CACHED_PASS_CHUNKS : DLAGP_MAX_PASS_CHUNKS;
    constexpr int A_STRIDE = 4 * PASS_CHUNKS * DLAGP_CHUNK_STEPS + 2;  // 58 (66 cached): stride/2 odd -> conflict-free b64 reads
    // one array: the wave's own finalisation (FUSED) stages its block's sums over both
    __shared__ double wave_lds[16 * DLAGP_RAW_STRIDE + 16 * A_STRIDE] __attribute__((aligned(16)));
    double *const raw = wave_lds;
    double *const a_lds = wave_lds + 16 * DLAGP_RAW_STRIDE;
    static_assert((16 * DLAGP_RAW_STRIDE) % 2 == 0, "a_lds[] is read and filled 16 bytes at a time");
    __shared__ int pos_lds[64];
    constexpr int NPAIR_MAX = PATCH ? 48 : 1;  // (sample, line) pairs of a wave and absorber
    constexpr int NCAND = 7;                   // candidate entries per pair, see below
    __shared__ double patch_factor[(PATCH == 1 ? NPAIR_MAX : 1) * NCAND];  // exact profile value of a core entry, -1 elsewhere
    __shared__ int patch_first[NPAIR_MAX];     // first candidate entry of an active pair
    __shared__ int patch_sample[NPAIR_MAX];    // its sample (0..15)
    __shared__ int patch_hit[NPAIR_MAX];       // per pass: active pairs touching the pass
    extern __shared__ double prep_lds[];
    static_assert(!PATCH || (LINES == 3 && MODE != 2), "the core patch is written for the 3-line Voigt loop");
    static_assert(PATCH != 2 || MODE == 0, "several absorbers per sample never meet the profile cache's level 1");
    static_assert(!LLS || LINES == 3, "the Lyman-limit term is implemented in the 3-line sample loop");
    static_assert(!PAIR || (PATCH == 1 && LINES == 3 && !LLS && MODE != 2),
                  "pair launches: one absorber per sample on the 3-line kernel, cores repaired once per wave");
    typedef double v4d_a16 __attribute__((ext_vector_type(4), aligned(16)));

    // finalize raises it; a later launch of a heterogeneous batch's level leaves the flag of the earlier ones alone
    // (FUSED: the waves of this launch raise it themselves, so the host clears it in front of the launch)
    if (!FUSED && blockIdx.x == 0 && threadIdx.x == 0 && !(launch_flags & DLAGP_LAUNCH_KEEP_STATUS)) *status = 0;
#ifdef DLAGP_TRACE_BUILD
    const unsigned long long trace_start = __builtin_amdgcn_s_memrealtime();
#endif
    // Wave priorities.  The two waves of a SIMD share the FP64 pipe; an MFMA holds it for 64 cycles, a vector
    // instruction for 4, so with equal priorities the wave that is in its MFMA loop starves the one that is in a
    // vector phase, which then reaches its own MFMA loop late and leaves latencies uncovered.  Measured best
    // (A/B on one device, -5 % kernel time against equal priorities): Voigt loop 2, the latency-bound parts
    // (pass set-up, convolution) 1, the dependent weight arithmetic at the head of each step 3, MFMAs 0.
    __builtin_amdgcn_s_setprio(2);
    unsigned long long t_entry = 0;
    if (STAMPS) t_entry = stamp_now();
    const WorkItem &item = items[blockIdx.x];
    const int b = __builtin_amdgcn_readfirstlane(item.b);
    const SightlineDev &sl = sightlines[b];
    const long gb = sl.batch_index;  // this sightline's position in the caller's sample arrays
    const int split = __builtin_amdgcn_readfirstlane(item.part);
    // Ranks above 20: the sightline's columns -- vech(M_p M_p') and M_p -- are contracted in groups of 210 + 20, each
    // group by its own waves with this same body: the group's slice of the operand (below) and its own partial sums
    // (epilogue).  The Voigt / weights work is repeated per group; against the 2.4x (k = 32) to 9.3x (k = 64) MFMA work
    // of such a rank that is the cheaper price than a kernel of its own.  Always 0 for rank <= 20.
    const int colgroup = __builtin_amdgcn_readfirstlane(item.colgroup);
    const long group = __builtin_amdgcn_readfirstlane(item.group);
    const int lane = threadIdx.x;
    const int s_loc = lane & 15;
    const int slot = lane >> 4;

    // descriptor fields used in the loops, read once
    const int num_entries = sl.num_entries;
    const gptr_d lam_ptr = as_global<gptr_d>(sl.lam);
    // P and the per-pixel records are streamed with buffer loads: scalar resource + per-lane byte
    // offset + scalar step offset + immediate, i.e. no vector address arithmetic in the loops
    const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)as_global<gptr_d>(sl.Pm + (long)colgroup * sl.nsteps_m * (2 * DLAGP_MFMA_PAIRS * DLAGP_TILE_DOUBLES)), 0,
        sl.nsteps_m * (2 * DLAGP_MFMA_PAIRS * DLAGP_TILE_DOUBLES * 8), DLAGP_BUFFER_FLAGS);
    const __amdgpu_buffer_rsrc_t aux_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)as_global<gptr_d>(sl.aux), 0, sl.n_slots * (DLAGP_AUX_FIELDS * 8), DLAGP_BUFFER_FLAGS);
    const int p_voff = lane * 16;                         // one double2 per lane and tile pair
    const int rec_voff = slot * (DLAGP_AUX_FIELDS * 8);   // this lane's pixel within a step
    double tap[4];  // symmetric profile: taps 0..3 (tap[t] == tap[6 - t], checked in dlagp_create)
#pragma unroll
    for (int t = 0; t < 4; ++t) tap[t] = sl.taps[t];

    const long sample0 = group * DLAGP_MFMA_GROUP;  // first sample of this wave
    const int L = LINES > 0 ? LINES : sl.num_lines;
    const int prep_stride = LLS ? 6 : L + 1;  // LLS rows: {m'_0, m'_1, m'_2, N log2e, scaled 1 / limit, T}

    // per-sample constants of this wave -> LDS (rows beyond S repeat the last sample; never stored)
    const int wave_prep = CACHED ? 0 : 16 * kd * prep_stride;
    // Cached levels (MODE 2).  Sample s of level kd stacks the cached profiles of QMC point s and of the points
    // base[0][s] .. base[kd - 2][s] (reference dla_gp.py:135-150), multiplied in that order.  Row s is streamed from HBM
    // once per level; the stacked rows are few -- the resampled indices of a level concentrate on a handful of QMC points
    // (profiles/r03_experiments.txt item 16; per wave and level 1.1-1.3 distinct rows on catalogue data, 2.3-3.1 on the
    // bench's sightlines, profiles/r04_experiments.txt) -- so the wave lists the DISTINCT rows its 16 samples stack over
    // all kd - 1 levels (the "pool", <= 16), reads each of them once per pass, from L2, and every sample picks its rows
    // out of LDS.  No running product is kept in memory: a level costs ONE streamed row per sample and no write-back
    // (up to round 3: the running product read, multiplied by one row and written back -- 72 KB per sample and level).
    // A wave whose samples stack more than 16 distinct rows (flat likelihoods) reads them level by level, 16 rows at a
    // time, at the head of every pass (`pooled` = false; also forced by launch flag DLAGP_LAUNCH_GENERAL_ROWS).
    int *row_lds = reinterpret_cast<int *>(prep_lds);  // [kd - 1][16] cached row of (level, sample), as a byte offset; then [NPOOL] pool rows
    constexpr int MAXJ = DLAGP_MAX_ABSORBERS - 1;  // stacked rows per sample
    constexpr int POOLJ = 3;                       // ... that the pool serves (the reference sweeps up to 4 absorbers)
    int *pool_row_lds = row_lds + MAXJ * 16;
    unsigned long long slot_bits[POOLJ] = {};  // pool slot of (level, sample): 4 bits each, wave-uniform
    bool pooled = false;
    int pool_used = 0;  // slots of the pool in use
    if constexpr (CACHED) {
        int npool = 0;
        bool overflow = (launch_flags & DLAGP_LAUNCH_GENERAL_ROWS) != 0 || kd - 1 > POOLJ;
        int pool_val = 0;  // lane t < npool: row of pool slot t
        for (int j = 1; j < kd; ++j) {
            long smp = sample0 + (lane & 15);
            smp = smp < S ? smp : S - 1;
            const int mine = (int)sample_point(src, gb, smp, j, S, kd);
            if (lane < 16) row_lds[(j - 1) * 16 + lane] = (mine >> 4) * ((int)src.cache_group_stride * 8) + (mine & 15) * (DLAGP_CACHE_TILE_SLOTS * 8);
            for (int ss = 0; ss < 16; ++ss) {
                const int v = __builtin_amdgcn_readlane(mine, ss);
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(lane < npool && pool_val == v);
                int slot_id;
                if (hit) {
                    slot_id = __builtin_ctzll(hit);
                } else if (npool < NPOOL) {
                    slot_id = npool;
                    if (lane == npool) pool_val = v;
                    ++npool;
                } else {
                    slot_id = 0;
                    overflow = true;
                }
                if (j <= POOLJ) slot_bits[j - 1] |= (unsigned long long)slot_id << (4 * ss);
            }
        }
        pooled = !overflow;
        pool_used = pooled ? npool : 0;
        // pool slots beyond the rows in use repeat slot 0's row (the rows are read four at a time, issue_rows)
        const int first = __builtin_amdgcn_readfirstlane(pool_val);
        // kept as the byte offset of the row within tile 0 of its group (issue_rows)
        const int pool_point = lane < npool ? pool_val : first;
        if (lane < NPOOL)
            pool_row_lds[lane] = (pool_point >> 4) * ((int)src.cache_group_stride * 8) + (pool_point & 15) * (DLAGP_CACHE_TILE_SLOTS * 8);
        lds_wave_sync();
    }
    for (int i = lane; i < wave_prep; i += 64) {
        const int which = i % prep_stride;           // 0..L-1: line multiplier, L: column density
        const int sample_abs = i / prep_stride;      // sample * kd + absorber
        long smp = sample0 + sample_abs / kd;
        smp = smp < S ? smp : S - 1;
        const int j = sample_abs % kd;
        const double zj = sample_z(src, gb, smp, j, S, kd);
        double value;
        if (which < L) {
            value = line_multiplier(which, zj) * DLAGP_LINES.inv_sqrt2_sigma;
        } else if (which == L) {
            value = sample_nhi(src, gb, smp, j, S, kd) * 1.44269504088896338700e+00;
        } else if (which == L + 1) {
            // (lambda / (1 + z) / limit)^3 / 10^17.2 = (lambda * value)^3
            value = 1.8478497974222906e-06 /* 10^(-17.2/3) */ / ((1.0 + zj) * DLAGP_LYMAN_LIMIT_ANGSTROM);
        } else {
            // smallest double T with fl(T / (1 + z)) > limit: start at limit (1 + z) and step ulps
            const double zp1 = 1.0 + zj;
            double t = DLAGP_LYMAN_LIMIT_ANGSTROM * zp1;
            if (t > 0.0 && t < 1e300) {
                for (int it = 0; it < 8 && t / zp1 > DLAGP_LYMAN_LIMIT_ANGSTROM; ++it)
                    t = __longlong_as_double(__double_as_longlong(t) - 1);
                for (int it = 0; it < 16 && !(t / zp1 > DLAGP_LYMAN_LIMIT_ANGSTROM); ++it)
                    t = __longlong_as_double(__double_as_longlong(t) + 1);
            }
            value = t;
        }
        // a NaN redshift makes every line NaN, np.nansum drops them all and the profile is 1 (voigt.py:309):
        // without the in-loop NaN test that is a zero column density
        // (and a finite multiplier, so that the optical depth the zero multiplies is finite)
        // (with the Lyman-limit term the reference's profile is NaN instead: rest wavelength NaN, voigt_lls.py:273-281)
        if (PATCH && !(zj == zj)) value = LLS && which == L ? __builtin_nan("") : 0.0;
        prep_lds[i] = value;
    }

    const int pass_begin = __builtin_amdgcn_readfirstlane(item.pass_begin);
    const int pass_end = __builtin_amdgcn_readfirstlane(item.pass_end);

    // ---------------- core patch factors of this wave (PATCH) ----------------
    int npatch = 0;  // active (sample, line) pairs, wave-uniform
    // PATCH == 2 (several absorbers per sample): the first candidate entry of every (sample, absorber, line), behind
    // the per-sample constants in dynamic LDS; the cores are repaired pass by pass, absorber by absorber (below)
    int *centre_lds = reinterpret_cast<int *>(prep_lds + (DLAGP_MFMA_GROUP + 1) * kd * prep_stride);
    if constexpr (PATCH == 2) {
        lds_wave_sync();  // per-sample constants visible
        const double x_offset = -DLAGP_LINES.c * DLAGP_LINES.inv_sqrt2_sigma;
        for (int i = lane; i < 16 * kd * 3; i += 64) {
            const double m = prep_lds[(i / 3) * prep_stride + i % 3];
            const double e_c = (log10(-x_offset / m) - sl.grid_log0) * sl.grid_inv_dlog;
            centre_lds[i] = e_c > -16.0 && e_c < 1.0e9 ? (int)floor(e_c - 3.5) + 1 : -(1 << 30);
        }
        lds_wave_sync();
    }
    if constexpr (PATCH == 1) {
        lds_wave_sync();  // per-sample constants visible
        const double x_offset = -DLAGP_LINES.c * DLAGP_LINES.inv_sqrt2_sigma;
        const int first_entry = passes[pass_begin].entry_start;
        const int last_entry = passes[pass_end - 1].entry_start + DLAGP_PASS_ENTRIES - 1;
        // lane p < 48 owns pair (sample p / 3, line p % 3): centre of the line in entry units.
        // x = lambda m' + x_offset vanishes at lambda_c = -x_offset / m'; candidates are the 7 entries inside
        // (e_c - 3.5, e_c + 3.5): the core half-width is 16 / 5.37 = 2.98 entries at 1e-4 dex, and the grid
        // may deviate from uniformity by 0.4 entry (checked in dlagp_create).
        bool active = false;
        int cand0 = 0;
        if (lane < 48) {
            const double m = prep_lds[(lane / 3) * prep_stride + lane % 3];
            const double e_c = (log10(-x_offset / m) - sl.grid_log0) * sl.grid_inv_dlog;
            if (e_c > -16.0 && e_c < 1.0e9) {  // finite, and not absurdly far bluewards of the grid
                cand0 = (int)floor(e_c - 3.5) + 1;
                active = cand0 + NCAND - 1 >= first_entry && cand0 <= last_entry;
            }
        }
        const unsigned long long amask = __builtin_amdgcn_ballot_w64(active);
        npatch = __builtin_popcountll(amask);
        if (active) {
            const int rank = __builtin_popcountll(amask & ((1ull << lane) - 1));
            patch_first[rank] = cand0;
            patch_sample[rank] = lane / 3 | (lane % 3) << 8;
        }
        lds_wave_sync();
        // All the wavelengths of the candidate entries are fetched before the first is used: one trip to L2 for the wave
        // instead of one per round of 64 candidates (<= 48 pairs x 7 = 6 rounds, each with its table look-ups behind it).
        constexpr int ROUNDS = (NPAIR_MAX * NCAND + 63) / 64;
        const int ncand = npatch * NCAND;
        double lam_c[ROUNDS];
        int e_c[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int item = r * 64 + lane;
            const int rank = item < ncand ? item / NCAND : 0;
            e_c[r] = patch_first[rank] + item % NCAND;
            const int e_load = e_c[r] < 0 ? 0 : (e_c[r] < num_entries ? e_c[r] : num_entries - 1);
            lam_c[r] = r * 64 < ncand ? lam_ptr[e_load] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (r * 64 >= ncand) break;
            const int item = r * 64 + lane;
            if (item < ncand) {
                const int rank = item / NCAND;
                const int packed = patch_sample[rank];
                const int ss = packed & 255, l = packed >> 8;
                const int e = e_c[r];
                const double lam_e = lam_c[r];
                const double *mrow = prep_lds + ss * prep_stride;
                const double x = fma(lam_e, mrow[l], x_offset);  // the loop's expression, bit for bit
                double factor = -1.0;  // marks "not a core entry" (profile values are >= 0)
                if (e >= 0 && e < num_entries && !(x * x >= DLAGP_FAD_XSPLIT * DLAGP_FAD_XSPLIT)) {
                    const double core = DLAGP_LINES.k_core[l] * core_taylor(x, DLAGP_LINES.y[l]);
                    double tau = core == core ? core : 0.0;  // a NaN term is dropped (np.nansum, voigt.py:309)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const double xk = fma(lam_e, mrow[k], x_offset);
                        const double wing = wing_term(xk * xk, DLAGP_LINES.wing[k]);
                        if (k != l) tau += wing;  // Lyman lines are > 40 A apart: never two cores at one entry
                    }
                    if constexpr (LLS) {
                        const double rho = lam_e * mrow[4];
                        if (!(lam_e >= mrow[5])) tau += rho * rho * rho;  // voigt_lls.py:276-281
                    }
                    factor = fast_exp2(-mrow[3] * tau);
                }
                patch_factor[item] = factor;
            }
        }
        lds_wave_sync();
    }

    d4 acc[DLAGP_MFMA_NT];
#pragma unroll
    for (int t = 0; t < DLAGP_MFMA_NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    double acc_sw = 0.0, acc_sq = 0.0;  // the two 4x4x4 MFMAs: vech columns 208.. and M columns 16..
    double sum_r2 = 0.0;
    LogProduct dprod;

    // compile-time switch: the stamp accumulators would otherwise hold ~12 SGPRs in production too
    constexpr bool stamps = STAMPS;
    unsigned long long t_voigt = 0, t_conv = 0, t_mfma = 0, t_mark = 0, t_start = 0;
    if (stamps) t_start = stamp_now();

    // B operand: 8 x double2 per lane and step (tile pair j -> tiles 2j, 2j+1); per-pixel record
    // {y, mu | omega2, v}.  Inside a pass both are reloaded one step ahead of use; the first step
    // of a pass is loaded after the convolution so that the registers are free during phase 1.
    constexpr int NPAIR = DLAGP_MFMA_PAIRS;
    constexpr int NBIGPAIR = DLAGP_MFMA_NT / 2;
    constexpr int STEP_BYTES = 2 * DLAGP_MFMA_PAIRS * DLAGP_TILE_DOUBLES * 8;  // 8192
    constexpr int REC_STEP_BYTES = DLAGP_STEP_PIXELS * DLAGP_AUX_FIELDS * 8;  // 256
    v2d frag[NPAIR];
    v2d rec_ym, rec_ov;

#ifndef DLAGP_CACHE_STORE_AUX
#define DLAGP_CACHE_STORE_AUX 2  // cache policy of level 1's stores of the profiles: nt
#endif
#ifndef DLAGP_CACHE_LOAD_AUX
#define DLAGP_CACHE_LOAD_AUX 2  // cache policy of the streamed rows of a cached level: nt
#endif
#ifndef DLAGP_PRIO_VOIGT
#define DLAGP_PRIO_VOIGT 2
#endif
#ifndef DLAGP_PRIO_CONV
#define DLAGP_PRIO_CONV 1
#endif
#ifndef DLAGP_PRIO_HEAD
#define DLAGP_PRIO_HEAD 3
#endif
#ifndef DLAGP_PRIO_MFMA
#define DLAGP_PRIO_MFMA 0
#endif
#ifndef DLAGP_PREFETCH_PASS
#define DLAGP_PREFETCH_PASS 1
#endif
    // The descriptor and the wavelengths of a pass are fetched during the pass before it: read at the head of
    // their own pass they are two dependent trips to L2 before the Voigt loop can start.
    // cached levels: the item's chunk range [c_first, c_last) (an item's passes are consecutive chunk ranges,
    // plan_passes) in passes that are the 64-slot TILES of the cache (dlagp_device.h): the first and the last pass of an
    // item may cover part of their tile
    int c_first = 0, c_last = 0;
    if constexpr (CACHED) {
        c_first = __builtin_amdgcn_readfirstlane(passes[pass_begin].chunk_begin);
        c_last = __builtin_amdgcn_readfirstlane(passes[pass_end - 1].chunk_begin + passes[pass_end - 1].nchunks);
    }
    const int tile_first = c_first / PASS_CHUNKS;
    const int loop_begin = CACHED ? 0 : pass_begin;
    const int loop_end = CACHED ? (c_last + PASS_CHUNKS - 1) / PASS_CHUNKS - tile_first : pass_end;
    auto pass_desc = [&](int p) __attribute__((always_inline)) -> PassDesc {
        if constexpr (CACHED) {
            const int tile_chunk = (tile_first + p) * PASS_CHUNKS;
            const int lo = tile_chunk > c_first ? tile_chunk : c_first;
            const int hi = tile_chunk + PASS_CHUNKS < c_last ? tile_chunk + PASS_CHUNKS : c_last;
            return PassDesc{0, lo, hi - lo, 0};
        } else {
            return passes[p];
        }
    };
    PassDesc pd_next = pass_desc(loop_begin);
    double lam_next = 0.0;
    if (DLAGP_PREFETCH_PASS && !CACHED) {
        int e = __builtin_amdgcn_readfirstlane(pd_next.entry_start) + lane;
        e = e < num_entries ? e : num_entries - 1;
        lam_next = lam_ptr[e];
    }
    // cached levels (MODE 2): per sample the cached row of its own QMC point (streamed), and the wave's pool of stacked
    // rows, for the 64 pixel slots of a pass; loaded a pass ahead, see the step loop
    double rowreg[CACHED ? 16 : 1];  // only the level-by-level path of a wave beyond the pool
    // Row reads are raw buffer loads: ONE scalar resource over the sightline's part of the cache, the (tile, row) as a
    // scalar byte offset, the slot within the tile as the per-lane offset -- no 64-bit vector addresses (32 of them at a
    // time spilled a hundred registers).  The host keeps a sightline's part of the cache below 2^31 bytes.
    // Layout (dlagp_device.h): [group of 16 points][tile of 64 slots][16 points][64 slots] -- the 16 own rows of a
    // wave's pass are ONE contiguous 8 KB block and its passes follow each other in memory (round 4; with rows of
    // n_slots doubles a pass read sixteen 512-byte pieces 24 KB apart).
    const int group_bytes = (int)src.cache_group_stride * 8;
    const long cache_groups = (S + DLAGP_MFMA_GROUP - 1) / DLAGP_MFMA_GROUP;
    const __amdgpu_buffer_rsrc_t cache_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)as_global<gptr_d>(CACHED ? src.cache + (long)b * cache_groups * src.cache_group_stride : nullptr), 0,
        CACHED ? (int)(cache_groups * src.cache_group_stride * 8) : 0, DLAGP_BUFFER_FLAGS);
    constexpr int TILE_BYTES = DLAGP_MFMA_GROUP * DLAGP_CACHE_TILE_SLOTS * 8;  // 8192
    constexpr int TILE_ROW_BYTES = DLAGP_CACHE_TILE_SLOTS * 8;                 // 512
    // The 32 rows of a pass go from memory straight into LDS (`buffer_load_dwordx4 ... lds`: 32 lanes x 16 bytes = the 64
    // slots of one row per instruction) -- sample ss's own row into a_lds[ss][], where its product will stand, pool row t
    // into raw[t][] (unused on these levels) -- so rows in flight hold no registers: up to round 4 they were 64 of the
    // wave's 253.  Always the whole tile, whatever part of it the pass covers.
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto issue_rows = [&](const PassDesc &pd) __attribute__((always_inline)) {
        if constexpr (CACHED) {
            const int tile_off = __builtin_amdgcn_readfirstlane(pd.chunk_begin / PASS_CHUNKS) * TILE_BYTES;
            const int own_off = (int)group * group_bytes + tile_off;
            const int voff = (lane & 31) * 16;
            const long last = S - 1 - sample0;  // rows beyond S repeat the last sample (never stored)
            int pool_off[NPOOL];
#pragma unroll
            for (int t = 0; t < NPOOL; ++t) pool_off[t] = __builtin_amdgcn_readfirstlane(pool_row_lds[t]) + tile_off;
            // a_lds[] was last read by the step before (its `a` of the pass's last step): landed before any row does
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < 32) {
#pragma unroll
                for (int ss = 0; ss < 16; ++ss) {
                    const int rel = (int)(ss < last ? ss : last);
                    // read once per level: streamed (nt) past the L2 that serves the sightline's operand and the pool's rows
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(cache_rsrc, (lds_ptr)(a_lds + ss * A_STRIDE), 16, voff,
                                                             own_off + rel * TILE_ROW_BYTES, 0, DLAGP_CACHE_LOAD_AUX);
                }
                // only the pool's rows, four at a time (the slots of a group beyond the pool repeat slot 0's row): these
                // are the hot lines of the sightline -- every wave of it reads the same few rows at about the same tile
#pragma unroll
                for (int t0 = 0; t0 < NPOOL; t0 += 4)
                    if (t0 < pool_used) {
#pragma unroll
                        for (int t = t0; t < t0 + 4; ++t)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(cache_rsrc, (lds_ptr)(raw + t * 64), 16, voff, pool_off[t], 0, 0);
                    }
            }
        }
    };
    int item_last_step = 0;  // cached levels: where the rolling operand prefetch stops
    if constexpr (CACHED) {
        issue_rows(pd_next);  // the first pass's rows
        __builtin_amdgcn_sched_barrier(0);
        // ... and the operands of the item's first step, in the step loop's order (see there)
        const int step = DLAGP_CHUNK_STEPS * c_first;
        const int soff = step * STEP_BYTES;
        rec_ym = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, rec_voff, step * REC_STEP_BYTES, 0));
        rec_ov = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, rec_voff + 16, step * REC_STEP_BYTES, 0));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) {
            frag[j] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(
                                                  p_rsrc, p_voff + (j & 3) * 1024, soff + (j >> 2) * 4096, 0));
            __builtin_amdgcn_sched_barrier(0);
        }
        item_last_step = DLAGP_CHUNK_STEPS * c_last - 1;
    }
    for (int pass = loop_begin; pass < loop_end; ++pass) {
        const PassDesc pd_v = DLAGP_PREFETCH_PASS ? pd_next : pass_desc(pass);
        const int entry_start = __builtin_amdgcn_readfirstlane(pd_v.entry_start);
        // Steps of the pass: operand steps step0 + st, slots 4 st .. 4 st + 3 of a_lds[], st = st_lo .. pass_steps - 1.
        // Voigt passes start at their first chunk (st_lo = 0); a cached pass is a tile of the cache, of which the item's
        // first pass may start (and its last end) inside.
        const int st_lo = CACHED ? __builtin_amdgcn_readfirstlane(DLAGP_CHUNK_STEPS * (pd_v.chunk_begin % PASS_CHUNKS)) : 0;
        const int step0 = __builtin_amdgcn_readfirstlane(DLAGP_CHUNK_STEPS * pd_v.chunk_begin) - st_lo;
        const int pass_steps = st_lo + __builtin_amdgcn_readfirstlane(DLAGP_CHUNK_STEPS * pd_v.nchunks);
        if (DLAGP_PREFETCH_PASS) pd_next = pass_desc(pass + 1 < loop_end ? pass + 1 : pass);

        // window start of pixel (lane) of this pass, relative to the pass; consumed after the Voigt loop
        const int my_rel = lane < pass_steps * DLAGP_STEP_PIXELS ? lane : 0;
        const double my_pos_f = __builtin_bit_cast(
            double, __builtin_amdgcn_raw_buffer_load_b64(aux_rsrc, my_rel * (DLAGP_AUX_FIELDS * 8) + 32,
                                                         step0 * REC_STEP_BYTES, 0));

        if constexpr (CACHED) {
            // ---------------- phase 1 (cached): the product of the sample's stacked cached profiles ----------------
            // The rows of this pass were sent on their way -- memory to LDS -- between the last two steps of the pass
            // before it (below), so their trip to HBM / L2 ran beside that step's MFMAs; what is left of it is hidden behind
            // the other resident wave.  Same multiplication order as the recomputing kernel: ((c_s c_b0) c_b1) c_b2 --
            // bit-identical.
            if (stamps) t_mark = stamp_now();
            double *pool_lds = raw;  // [NPOOL][64]; raw[] is otherwise unused on the cached levels
            // the rows have landed in LDS; the operand loads of the step that followed them -- two record loads and NPAIR
            // tile pairs, issued behind the rows both in front of the pass loop and in the last step of every pass
            // (DLAGP_STEP_BODY with FETCH) -- stay in flight.  The count is tied to the step body's loads here: a step
            // that fetched more (or fewer) would make the products read a_lds[] and the pool before the DMA has landed.
            constexpr int kOperandLoadsPerStep = 2 + NPAIR;
            static_assert(kOperandLoadsPerStep == 10 && NPAIR == DLAGP_MFMA_PAIRS,
                          "the step body issues 2 record loads + NPAIR operand loads: keep this wait count in step with it");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kOperandLoadsPerStep) : "memory");
            __builtin_amdgcn_wave_barrier();
            // the lane part of the LDS addresses is made opaque per pass, or the compiler keeps all (slot, lane) addresses
            // in registers across the pass loop (44 spills)
            int lane_off = lane * 8;
            asm volatile("" : "+v"(lane_off));
            if (pooled) {
                // Every sample picks its kd - 1 rows out of the pool: reads of four samples in flight, then their
                // products, written over the sample's own row.  The slot numbers are wave-uniform scalars.
                const char *pool_bytes = reinterpret_cast<const char *>(pool_lds);
                char *a_bytes = reinterpret_cast<char *>(a_lds);
                auto stack = [&](auto depth) __attribute__((always_inline)) {
                    constexpr int D = decltype(depth)::value;  // stacked rows per sample = kd - 1
#pragma unroll
                    for (int s0 = 0; s0 < 16; s0 += 4) {
                        double r[4][D], o[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            o[u] = *reinterpret_cast<const double *>(a_bytes + ((s0 + u) * (A_STRIDE * 8) + lane_off));
#pragma unroll
                            for (int j = 0; j < D; ++j) {
                                const int slot_off = (int)((slot_bits[j] >> (4 * (s0 + u))) & 15) * 512;
                                r[u][j] = *reinterpret_cast<const double *>(pool_bytes + (slot_off + lane_off));
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            double prod = o[u];
#pragma unroll
                            for (int j = 0; j < D; ++j) prod *= r[u][j];
                            *reinterpret_cast<double *>(a_bytes + ((s0 + u) * (A_STRIDE * 8) + lane_off)) = prod;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                if (kd == 2) stack(std::integral_constant<int, 1>());
                else if (kd == 3) stack(std::integral_constant<int, 2>());
                else stack(std::integral_constant<int, 3>());
            } else {
                // more than NPOOL distinct rows in this wave: the stacked rows level by level, 16 reads at a time
                const int tile_off = (step0 / (DLAGP_CHUNK_STEPS * PASS_CHUNKS)) * TILE_BYTES;
                char *a_bytes = reinterpret_cast<char *>(a_lds);
                for (int j = 1; j < kd; ++j) {
#pragma unroll
                    for (int ss = 0; ss < 16; ++ss)
                        rowreg[ss] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                                                                    cache_rsrc, lane_off, __builtin_amdgcn_readfirstlane(row_lds[(j - 1) * 16 + ss]) + tile_off, 0));
#pragma unroll
                    for (int ss = 0; ss < 16; ++ss) {
                        double *dst = reinterpret_cast<double *>(a_bytes + (ss * (A_STRIDE * 8) + lane_off));
                        *dst = *dst * rowreg[ss];
                    }
                }
            }
            lds_wave_sync();
            if (stamps) t_conv += stamp_now() - t_mark;
        } else {
        // ---------------- phase 1: raw profile and convolution, absorber by absorber ----------------
            double lam;
            if (DLAGP_PREFETCH_PASS) {
                lam = lam_next;
                int e = __builtin_amdgcn_readfirstlane(pd_next.entry_start) + lane;  // the next pass's (or, last, this one's)
                e = e < num_entries ? e : num_entries - 1;
                lam_next = lam_ptr[e];
            } else {
                int e = entry_start + lane;
                e = e < num_entries ? e : num_entries - 1;
                lam = lam_ptr[e];
            }
            for (int jd = 0; jd < kd; ++jd) {
                if (stamps) t_mark = stamp_now();
                if (LINES == 3) {
                    // the sample's four constants {m'_0, m'_1, m'_2, N log2e} are one 32-byte LDS broadcast
                    // read, fetched one sample ahead so that the loop never waits on it
                    const v4d *sp4 = reinterpret_cast<const v4d *>(prep_lds) + jd;
                    // The three line multipliers of the next sample are read right after their last use in this
                    // one, into the same registers (no copies); the column density, needed until the end of the
                    // body, rotates through one extra register pair.
                    const double *lls_row = prep_lds + jd * 6;  // LLS rows: 6 doubles per (sample, absorber)
                    const v4d first = LLS ? (v4d)*reinterpret_cast<const v4d_a16 *>(lls_row) : sp4[0];
                    double m0 = first.x, m1 = first.y, m2 = first.z, ncol = first.w;
                    double lls_scale = 0.0, lls_edge = 0.0;
                    if constexpr (LLS) {
                        lls_scale = lls_row[4];
                        lls_edge = lls_row[5];
                    }
                    const double x_offset = -DLAGP_LINES.c * DLAGP_LINES.inv_sqrt2_sigma;
                    const double c8v = vgpr_constant(DLAGP_LINES.exp2c[8]);
                    const double w3v[3] = {vgpr_constant(DLAGP_LINES.wing[0][3]), vgpr_constant(DLAGP_LINES.wing[1][3]),
                                           vgpr_constant(DLAGP_LINES.wing[2][3])};
                    __builtin_amdgcn_s_setprio(DLAGP_PRIO_VOIGT);
                    // one running LDS address (hipcc otherwise rebuilds lane * 8 + offset for every sample)
                    __attribute__((address_space(3))) double *raw_w = (__attribute__((address_space(3))) double *)raw + lane;
                    asm volatile("" : "+v"(raw_w));
                    if constexpr (PAIR) {
                        // rows ss and ss + 8: one QMC point, two column densities.  The optical depth per unit column -- the
                        // wing polynomials and the shared reciprocal, three quarters of the body -- is formed once.
#pragma unroll 1
                        for (int ss = 0; ss < 8; ++ss) {
                            const double xs[3] = {fma(lam, m0, x_offset), fma(lam, m1, x_offset), fma(lam, m2, x_offset)};
                            __builtin_amdgcn_sched_barrier(0);
                            const v4d nxt = sp4[(ss + 1) * kd];
                            const double ncol_b = reinterpret_cast<const double *>(sp4 + (ss + 8) * kd)[3];
                            m0 = nxt.x;
                            m1 = nxt.y;
                            m2 = nxt.z;
                            __builtin_amdgcn_sched_barrier(0);
                            const double tau = optical_depth<LINES, false>(lam, nullptr, 3, w3v, xs);
                            raw_w[0] = fast_exp2_table(-ncol * tau, DLAGP_LINES.exp2c, c8v);                        // voigt.py:309
                            raw_w[8 * DLAGP_RAW_STRIDE] = fast_exp2_table(-ncol_b * tau, DLAGP_LINES.exp2c, c8v);
                            raw_w += DLAGP_RAW_STRIDE;
                            ncol = nxt.w;
                        }
                    } else
#pragma unroll 1
                    for (int ss = 0; ss < 16; ++ss) {
                        const double xs[3] = {fma(lam, m0, x_offset), fma(lam, m1, x_offset), fma(lam, m2, x_offset)};
                        __builtin_amdgcn_sched_barrier(0);
                        // the last read lands in the spare 17th row
                        const v4d nxt = LLS ? (v4d)*reinterpret_cast<const v4d_a16 *>(lls_row + (ss + 1) * kd * 6) : sp4[(ss + 1) * kd];
                        m0 = nxt.x;
                        m1 = nxt.y;
                        m2 = nxt.z;
                        __builtin_amdgcn_sched_barrier(0);
                        double tau = optical_depth<LINES, PATCH == 0>(lam, nullptr, 3, w3v, xs);
                        if constexpr (LLS) {
                            const double rho = lam * lls_scale;
                            tau += lam >= lls_edge ? 0.0 : rho * rho * rho;  // voigt_lls.py:276-281
                            lls_scale = lls_row[(ss + 1) * kd * 6 + 4];
                            lls_edge = lls_row[(ss + 1) * kd * 6 + 5];
                        }
                        *raw_w = fast_exp2_table(-ncol * tau, DLAGP_LINES.exp2c, c8v);  // voigt.py:309
                        raw_w += DLAGP_RAW_STRIDE;
                        ncol = nxt.w;
                    }
                } else {
#pragma unroll 2
                    for (int ss = 0; ss < 16; ++ss) {
                        const double *sp = prep_lds + (ss * kd + jd) * prep_stride;
                        const double tau = optical_depth<LINES>(lam, sp, L);
                        raw[ss * DLAGP_RAW_STRIDE + lane] = fast_exp2(-sp[L] * tau);
                    }
                }
                if (stamps) {
                    const unsigned long long now = stamp_now();
                    t_voigt += now - t_mark;
                    t_mark = now;
                }
                if (jd == 0) pos_lds[lane] = (int)my_pos_f - entry_start;
                lds_wave_sync();
                if constexpr (PATCH == 2) {
                    // line cores of absorber jd inside this pass: lane p < 48 owns (sample p / 3, line p % 3); the
                    // few pairs whose 7 candidate entries touch the pass get their core entries' exact profile
                    // value -- core form for that line, wing form for the other two -- written over raw[]
                    const int mine = lane < 48 ? centre_lds[((lane / 3) * kd + jd) * 3 + lane % 3] : -(1 << 30);
                    const bool hit = mine + NCAND - 1 >= entry_start && mine < entry_start + DLAGP_PASS_ENTRIES;
                    const unsigned long long hmask = __builtin_amdgcn_ballot_w64(hit);
                    if (hmask) {
                        if (hit) patch_hit[__builtin_popcountll(hmask & ((1ull << lane) - 1))] = lane;
                        lds_wave_sync();
                        const double x_offset = -DLAGP_LINES.c * DLAGP_LINES.inv_sqrt2_sigma;
                        const int nitems = __builtin_popcountll(hmask) * NCAND;
                        for (int base = 0; base < nitems; base += 64) {
                            const int item = base + lane;
                            const bool live = item < nitems;
                            const int pair = patch_hit[live ? item / NCAND : 0], j = item % NCAND;
                            const int ss = pair / 3, l = pair % 3;
                            const int rel = centre_lds[(ss * kd + jd) * 3 + l] + j - entry_start;
                            const bool inside = live && rel >= 0 && rel < DLAGP_PASS_ENTRIES;
                            const double lam_e = __shfl(lam, inside ? rel : 0);  // lane rel holds this entry's wavelength
                            if (inside && entry_start + rel < num_entries) {
                                const double *mrow = prep_lds + (ss * kd + jd) * prep_stride;
                                const double x = fma(lam_e, mrow[l], x_offset);  // the loop's expression, bit for bit
                                if (!(x * x >= DLAGP_FAD_XSPLIT * DLAGP_FAD_XSPLIT)) {
                                    const double core = DLAGP_LINES.k_core[l] * core_taylor(x, DLAGP_LINES.y[l]);
                                    double tau = core == core ? core : 0.0;  // a NaN term is dropped (np.nansum, voigt.py:309)
#pragma unroll
                                    for (int k = 0; k < 3; ++k) {
                                        const double xk = fma(lam_e, mrow[k], x_offset);
                                        const double wing = wing_term(xk * xk, DLAGP_LINES.wing[k]);
                                        if (k != l) tau += wing;  // Lyman lines are > 40 A apart: never two cores at one entry
                                    }
                                    if constexpr (LLS) {
                                        const double rho = lam_e * mrow[4];
                                        if (!(lam_e >= mrow[5])) tau += rho * rho * rho;  // voigt_lls.py:276-281
                                    }
                                    raw[ss * DLAGP_RAW_STRIDE + rel] = fast_exp2(-mrow[3] * tau);
                                }
                            }
                        }
                        lds_wave_sync();
                    }
                }
                if constexpr (PATCH == 1) {
                    // copy the core values that fall into this pass's 64 entries over raw[] (see the kernel header)
                    const int mine = patch_first[lane < npatch ? lane : 0];
                    const bool hit = lane < npatch && mine + NCAND - 1 >= entry_start && mine < entry_start + DLAGP_PASS_ENTRIES;
                    const unsigned long long hmask = __builtin_amdgcn_ballot_w64(hit);
                    if (hmask) {
                        if (hit) patch_hit[__builtin_popcountll(hmask & ((1ull << lane) - 1))] = lane;
                        lds_wave_sync();
                        const int nitems = __builtin_popcountll(hmask) * NCAND;
                        for (int base = 0; base < nitems; base += 64) {
                            const int item = base + lane;
                            if (item < nitems) {
                                const int rank = patch_hit[item / NCAND], j = item % NCAND;
                                const int rel = patch_first[rank] + j - entry_start;
                                const double value = patch_factor[rank * NCAND + j];
                                if (rel >= 0 && rel < DLAGP_PASS_ENTRIES && !(value < 0.0))
                                    raw[(patch_sample[rank] & 255) * DLAGP_RAW_STRIDE + rel] = value;
                            }
                        }
                        lds_wave_sync();
                    }
                }
                __builtin_amdgcn_s_setprio(DLAGP_PRIO_CONV);
                // convolution (voigt.py:319) on the kept pixels (dla_gp.py:388), product over absorbers.
                // Always all 14 steps, branch-free, so that the LDS reads of different steps overlap:
                // lanes beyond the pass's pixels re-read a valid window and write unused slots.
                // The reads of four steps are issued together (scheduling barriers keep hipcc from
                // serialising them one ds_read + wait at a time), then the arithmetic runs on them.
                const double *raw_row = raw + s_loc * DLAGP_RAW_STRIDE;
                double *a_dst = &a_lds[s_loc * A_STRIDE + slot];
                // (7 with several absorbers per sample -- their running product is one more register pair per step -- spills)
                constexpr int CONV_GROUP = PATCH == 1 && MODE == 0 ? 7 : 4;  // reads of that many steps in flight; the single-absorber kernel has the registers for 7
                constexpr int NW = BROADEN ? DLAGP_NUM_TAPS : 1;
#pragma unroll
                for (int g0 = 0; g0 < DLAGP_MAX_PASS_STEPS; g0 += CONV_GROUP) {
                    double wv[CONV_GROUP][NW];
                    double prev[CONV_GROUP];
#pragma unroll
                    for (int u = 0; u < CONV_GROUP; ++u) {
                        const int st = g0 + u;
                        if (st < DLAGP_MAX_PASS_STEPS) {
                            const double *win = raw_row + pos_lds[st * DLAGP_STEP_PIXELS + slot];
#pragma unroll
                            for (int t = 0; t < NW; ++t) wv[u][t] = win[t];
                            if (PATCH != 1) prev[u] = a_dst[st * DLAGP_STEP_PIXELS];  // PATCH == 1 implies one absorber
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < CONV_GROUP; ++u) {
                        const int st = g0 + u;
                        if (st < DLAGP_MAX_PASS_STEPS) {
                            // the instrument profile is symmetric (voigt.py:214-224): 4 coefficients, not 7
                            double conv = wv[u][0];
                            if constexpr (BROADEN) {
                                conv = (wv[u][0] + wv[u][6]) * tap[0];
                                conv = fma(wv[u][1] + wv[u][5], tap[1], conv);
                                conv = fma(wv[u][2] + wv[u][4], tap[2], conv);
                                conv = fma(wv[u][3], tap[3], conv);
                            }
                            if constexpr (PATCH == 1) a_dst[st * DLAGP_STEP_PIXELS] = conv;
                            else a_dst[st * DLAGP_STEP_PIXELS] = jd == 0 ? conv : prev[u] * conv;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                lds_wave_sync();
                if (stamps) t_conv += stamp_now() - t_mark;
            }
        }
        if (stamps) t_mark = stamp_now();

        // operands of the pass's first step, issued in exactly the order the loop below issues them
        // (records first, then tile pairs 0..7): loads complete in order and the wait counts at the
        // loop header are the minimum over both incoming edges, so any other order here makes
        // every step wait for all eight of its predecessor's prefetches instead of the oldest.
        // (Cached levels: the prefetch rolls on across passes -- the last step of a pass fetches the first of the next,
        // see below -- and the very first step's operands are issued in front of the pass loop.)
        if constexpr (!CACHED) {
            const int soff = step0 * STEP_BYTES;
            rec_ym = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, rec_voff, step0 * REC_STEP_BYTES, 0));
            rec_ov = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, rec_voff + 16, step0 * REC_STEP_BYTES, 0));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NPAIR; ++j) {
                frag[j] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(
                                                      p_rsrc, p_voff + (j & 3) * 1024, soff + (j >> 2) * 4096, 0));
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---------------- phase 2: weights and MFMAs, step by step ----------------
        const double *a_row = &a_lds[s_loc * A_STRIDE + slot];
        double a = a_row[st_lo * DLAGP_STEP_PIXELS];
        // One step of 4 pixel slots.  FETCH: also fetch the next step's operands; READ_A: and its `a` (not in the last
        // step of a cached pass: a_lds[] is being refilled).  A macro, not a lambda: the step loop of the Voigt kernels
        // is tuned to the instruction (an inlined lambda came out 2.6 % slower on configs[1]).
#define DLAGP_STEP_BODY(st, FETCH, READ_A)                                                                                     \
        {                                                                                                                \
            __builtin_amdgcn_s_setprio(DLAGP_PRIO_HEAD);                                                                 \
            const double a2 = a * a;                                                                                     \
            const double d = fma(rec_ov.x, a2, rec_ov.y);                                                                \
            const double h = fast_rcp1(d);                                                                               \
            const double r = fma(-rec_ym.y, a, rec_ym.x);                                                                \
            const double rh = r * h;                                                                                     \
            const double w = a2 * h;                                                                                     \
            const double q = a * rh;                                                                                     \
            sum_r2 = fma(r, rh, sum_r2);                                                                                 \
            dprod.multiply(d);                                                                                           \
            /* next step's operands; the last step of a pass re-reads its own (harmless, never used) -- on the cached */ \
            /* levels the first step of the next pass: an item's passes are consecutive steps of the operand */          \
            const int nst = (st) + 1 < pass_steps ? (st) + 1 : (st);                                                     \
            int nxt = step0 + nst;                                                                                       \
            if constexpr (CACHED) nxt = step0 + (st) + 1 < item_last_step ? step0 + (st) + 1 : item_last_step;          \
            if (FETCH) {                                                                                                 \
                rec_ym = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, rec_voff, nxt * REC_STEP_BYTES, 0)); \
                rec_ov = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, rec_voff + 16, nxt * REC_STEP_BYTES, 0)); \
                if (READ_A) a = a_row[nst * DLAGP_STEP_PIXELS];                                                          \
            }                                                                                                            \
            /* keep the prefetches above in front of the MFMAs: their results are not needed before */                   \
            /* the next step, 16 MFMAs (>= 1024 cycles) from here */                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            __builtin_amdgcn_s_setprio(DLAGP_PRIO_MFMA);                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            const int soff = nxt * STEP_BYTES;                                                                           \
            _Pragma("unroll") for (int j = 0; j < NPAIR; ++j) {                                                          \
                const v2d f = frag[j];                                                                                   \
                if (j < NBIGPAIR) {                                                                                      \
                    acc[2 * j] = __builtin_amdgcn_mfma_f64_16x16x4f64(2 * j < DLAGP_MFMA_NTW ? w : q, f.x, acc[2 * j], 0, 0, 0); \
                    acc[2 * j + 1] =                                                                                     \
                        __builtin_amdgcn_mfma_f64_16x16x4f64(2 * j + 1 < DLAGP_MFMA_NTW ? w : q, f.y, acc[2 * j + 1], 0, 0, 0); \
                } else {                                                                                                 \
                    /* A operand lane = pixel * 16 + sample is also the 4-block form's (sample = block * 4 + row) */     \
                    acc_sw = __builtin_amdgcn_mfma_f64_4x4x4f64(w, f.x, acc_sw, 0, 0, 0);                                \
                    acc_sq = __builtin_amdgcn_mfma_f64_4x4x4f64(q, f.y, acc_sq, 0, 0, 0);                                \
                }                                                                                                        \
                /* rolling prefetch: this register pair is next used one full step later */                              \
                if (FETCH)                                                                                               \
                    frag[j] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(                             \
                                                          p_rsrc, p_voff + (j & 3) * 1024, soff + (j >> 2) * 4096, 0));  \
            }                                                                                                            \
            /* pin the interleaving: two MFMAs, then the load that refills their operand pair */                         \
            _Pragma("unroll") for (int j = 0; j < NPAIR; ++j) {                                                          \
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); /* MFMA */                                            \
                if (FETCH) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); /* VMEM read */                            \
            }                                                                                                            \
        }
        if constexpr (MODE == 1) {
            // Level 1 of a cached sweep stores the pass's profiles right after the step loop (below).  The last step's
            // prefetch -- it "re-reads its own", never used -- would still be in flight there, in registers the store
            // block's LDS reads are given next: a wait for ten L2 trips per pass.  So the last step is written out without
            // the prefetch (-0.9 %; what level 1 still costs over the kernel without the store is mostly the clock,
            // profiles/r04_experiments.txt items 9, 10).
            for (int st = st_lo; st + 1 < pass_steps; ++st) DLAGP_STEP_BODY(st, true, true)
            DLAGP_STEP_BODY(pass_steps - 1, false, false)
        } else if constexpr (!CACHED) {
            for (int st = st_lo; st < pass_steps; ++st) DLAGP_STEP_BODY(st, true, true)
        } else {
            // Cached levels: the last two steps are written out so that the memory traffic of the NEXT pass can start
            // between them.  Vector-memory operations complete in order and the step loop waits for its operand
            // prefetches, so anything issued in front of it (the old place of the row reads) holds
            // up its first steps for a trip to HBM; issued here, behind the last operand prefetch of the pass, the
            // row reads of the next pass are in flight during a step of
            // MFMAs and nothing waits for them before the next pass's products.  (Inside the loop, behind a condition,
            // the compiler's wait counts at the loop head would have to assume the worst of both paths.)
            for (int st = st_lo; st + 2 < pass_steps; ++st) DLAGP_STEP_BODY(st, true, true)
            DLAGP_STEP_BODY(pass_steps - 2, true, true)  // a pass holds an even number of steps >= 2 (chunks of 8 slots)
            // Behind this condition the wait counts of the step after the join are the minimum over the two arms -- the
            // arm without reads -- so the last step waits for most of the row reads (vmcnt(8) in the ISA).  With the
            // reads unconditional the counts are exact (the step runs beside the reads) and nothing changes: 8208 / 8245 us
            // against 8205 / 8249 us on config 5 (profiles/r04_experiments.txt item 3) -- the rows' trip is waited for at
            // the head of the next pass instead.
            // (the products of the next pass wait with an explicit count -- the ten operand loads of the last step may
            // stay in flight -- so nothing may move across the rows' issue)
            __builtin_amdgcn_sched_barrier(0);
            if (pass + 1 < loop_end) issue_rows(pd_next);
            __builtin_amdgcn_sched_barrier(0);
            // the last step's prefetch is the next pass's first step: issued behind the rows, it is back long before
            // the products of the next pass are done, which up to round 4 were followed by a trip to L2 for it
            DLAGP_STEP_BODY(pass_steps - 1, true, false)
        }
#undef DLAGP_STEP_BODY
        // Level 1 of a multi-level sweep: keep the convolved profiles for the deeper levels.  Stored here, at the end of
        // the pass, not right after the convolution: vector-memory operations complete in order, so stores issued
        // in front of the step loop made its first steps wait for 16 write acknowledgements from HBM (MODE 1 ran 19 %
        // slower than MODE 0); from here they drain during the next pass's Voigt loop.  a_lds[] is intact until the
        // next convolution.
        if constexpr (MODE == 1) {
            const int npx = pass_steps * DLAGP_STEP_PIXELS;
            // 16 bytes per lane: lane = (h, slot pair j) stores slots 2 j, 2 j + 1 of the samples 2 i + h -- stores are
            // issue-bound (profiles/r03_experiments.txt): 8 instructions per pass instead of 16.  (The same for the
            // cached levels' row reads and write-backs spilled registers and ran 7 % slower: they keep 8 bytes per lane.)
            const int h = lane >> 5, j2 = 2 * (lane & 31);
            // tiled layout of the cache (dlagp_device.h): [group][tile of 64 slots][16 points][64 slots]; a slot pair
            // never straddles a tile (passes start on multiples of 8 slots)
            const int x = step0 * DLAGP_STEP_PIXELS + j2;
            const long cache_groups_1 = (S + DLAGP_MFMA_GROUP - 1) / DLAGP_MFMA_GROUP;
            // raw buffer stores: one scalar resource over the sightline's part of the cache (below 2^31 bytes, as the
            // cached levels need it), the group as the scalar offset, (tile, slot) as the lane's
            const __amdgpu_buffer_rsrc_t store_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void *)as_global<gptr_d>(src.cache + (long)b * cache_groups_1 * src.cache_group_stride), 0,
                (int)(cache_groups_1 * src.cache_group_stride * 8), DLAGP_BUFFER_FLAGS);
            const int group_off = (int)group * ((int)src.cache_group_stride * 8);
            const int lane_store = ((x / DLAGP_CACHE_TILE_SLOTS) * (DLAGP_MFMA_GROUP * DLAGP_CACHE_TILE_SLOTS) + (x % DLAGP_CACHE_TILE_SLOTS)) * 8;
            // streaming stores: 24 KB per sample written once and read a level later -- kept out of the L2 that
            // serves the sightline's operand.  All eight rows are read into registers of their own before the first
            // store: with one register set the compiler waited, before each LDS read, for the store before it to have
            // COMPLETED (no early release of a store's data registers on this target: `s_waitcnt vmcnt`) -- eight trips
            // to HBM one after the other per pass (-0.7 % alone, more beside other streams' kernels: the catalogue
            // driver).  Rows beyond S (the last group) are written too:
            // the cache holds whole groups of 16 rows and the cached levels never read those.
            if constexpr (PAIR) {
                // pair launch: only rows 8..15 are the DLA model's profiles -- points 8 group .. 8 group + 7, i.e. rows
                // 8 (group & 1) .. of the cache's group (group >> 1) of 16 points; four stores per lane
                const long cache_groups_p = (src.pair_S + DLAGP_MFMA_GROUP - 1) / DLAGP_MFMA_GROUP;
                const __amdgpu_buffer_rsrc_t pair_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)as_global<gptr_d>(src.cache + (long)b * cache_groups_p * src.cache_group_stride), 0,
                    (int)(cache_groups_p * src.cache_group_stride * 8), DLAGP_BUFFER_FLAGS);
                const int pair_group_off = (int)(group >> 1) * ((int)src.cache_group_stride * 8);
                const int half = (int)(group & 1) * 8;
                if (j2 < npx) {
                    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
                    v2d rows[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) rows[i] = *reinterpret_cast<const v2d *>(&a_lds[(8 + 2 * i + h) * A_STRIDE + j2]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, rows[i]), pair_rsrc,
                                                               lane_store + (half + 2 * i + h) * (DLAGP_CACHE_TILE_SLOTS * 8), pair_group_off,
                                                               DLAGP_CACHE_STORE_AUX);
                }
            } else
            if (j2 < npx) {
                typedef unsigned int u4 __attribute__((ext_vector_type(4)));
                v2d rows[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) rows[i] = *reinterpret_cast<const v2d *>(&a_lds[(2 * i + h) * A_STRIDE + j2]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, rows[i]), store_rsrc,
                                                           lane_store + (2 * i + h) * (DLAGP_CACHE_TILE_SLOTS * 8), group_off, DLAGP_CACHE_STORE_AUX);
            }
        }
        __builtin_amdgcn_s_setprio(DLAGP_PRIO_CONV);
        // the next pass overwrites raw[] / a_lds[]: order the LDS reads above before those writes
        lds_wave_sync();
        if (stamps) t_mfma += stamp_now() - t_mark;
    }
    if (stamps) t_mark = stamp_now();

    // ---------------- epilogue: partial sums of this (piece, block) for finalize_rank20<.., NATIVE> ----------------
    // The 14 full tiles go out as the MFMA leaves them (D layout: lane holds column lane & 15 of sample
    // (lane >> 4) + 4 r), two accumulator registers per store: 28 stores of 1 KB, 16 bytes per lane, no predicates (rows
    // beyond S are never read back as results) and no transposition -- the finalisation sorts them into its LDS image
    // while it sums the pieces (kernels_common.h, native_tile_offset).  Up to round 2 every tile went through LDS
    // first and left in 56 stores of 8 bytes per lane.
    // a (piece, block) of partial sums is DLAGP_MFMA_ROW_WIDTH x 16 doubles whatever the rank (= row_width at rank 20);
    // column group g's pieces follow those of group g - 1: [group][piece][block]
    if constexpr (FUSED) {
        // ---------------- epilogue (FUSED): this wave finishes its own block ----------------
        // Four rounds of four samples: round r is component r of every accumulator tile (the MFMA's D layout: lane holds
        // column lane & 15 of sample 4 r + (lane >> 4)), staged into LDS as [column][sample of the round] -- 233 columns x
        // stride 5 over raw[] / a_lds[], which nothing reads any more -- and factored by the SIXTEEN lanes of each DPP row
        // exactly as finalize_rank20<ROW16> factors it (factor_solve_row16: the same operations in the same order, and a
        // single piece has nothing to add up -- bit-identical to the two-kernel form, tests/test_gpu_parity.py).
        constexpr int FLS = 5, NC = DLAGP_FIN_COLS, K = DLAGP_FIN_K;
        static_assert(DLAGP_FIN_STAGED * FLS <= 16 * DLAGP_RAW_STRIDE + 16 * A_STRIDE, "the staged block must fit the wave's LDS");
        static_assert(K == DLAGP_MFMA_RANK && NC == DLAGP_MFMA_SMALL_W_COL + 2, "column layout of the rank-20 tiles");
        double logd = dprod.log_value();
        sum_r2 += __shfl_xor(sum_r2, 16);
        sum_r2 += __shfl_xor(sum_r2, 32);
        logd += __shfl_xor(logd, 16);
        logd += __shfl_xor(logd, 32);
        const double n_log_2pi = sl.n * DLAGP_LOG_2PI;
        double *const fin = wave_lds;
        static_for<0, 4>([&](auto r_c) __attribute__((always_inline)) {
            constexpr int r = decltype(r_c)::value;
            double *const mine = fin + slot;  // sample 4 r + slot of the block; slot = lane >> 4
#pragma unroll
            for (int t = 0; t < DLAGP_MFMA_NT; ++t)
                mine[((t < DLAGP_MFMA_NTW ? 16 * t : NC) + s_loc) * FLS] = acc[t][r];  // tile 13: u 0..15
            // 4x4x4 outputs: lane = row * 16 + block * 4 + column  ->  sample = block * 4 + row; this round is block r
            if (((lane >> 2) & 3) == r) {
                const int j = lane & 3;
                if (j < 2) mine[(DLAGP_MFMA_SMALL_W_COL + j) * FLS] = acc_sw;  // vech columns 208, 209
                mine[(NC + DLAGP_MFMA_SMALL_Q_COL + j) * FLS] = acc_sq;          // u 16..19
            }
            if (slot == 0 && (s_loc >> 2) == r) {  // after the shuffles every lane of sample s_loc holds its totals
                fin[(NC + K) * FLS + (s_loc & 3)] = sum_r2;
                fin[(NC + K + 1) * FLS + (s_loc & 3)] = logd;
            }
            lds_wave_sync();
            bool bad = false;
            double result = factor_solve_row16<FLS>(mine, s_loc, n_log_2pi, bad);
            // ordinary launch: sample sample0 + row; pair launch: point 8 group + (row & 7) of model row >> 3
            const int row = 4 * r + slot;
            const long smp = PAIR ? group * 8 + (row & 7) : sample0 + row;
            const long count = PAIR ? src.pair_S : S;
            if (s_loc == 4 && smp < count) {  // row 20 = t sits in lane 4 of the row
                if (bad) {
                    result = __builtin_nan("");
                    atomicOr(status, 1);
                }
                (PAIR && row >= 8 ? out2 : out)[gb * count + smp] = result;
            }
            lds_wave_sync();  // the next round's staging overwrites what this one read
        });
    } else {
    const int rw = DLAGP_MFMA_ROW_WIDTH;
    const long rows_left = S - sample0;  // valid samples of this wave
    const long ngroups = (S + DLAGP_MFMA_GROUP - 1) / DLAGP_MFMA_GROUP;
    const long nblocks = (long)nb * ngroups;
    __attribute__((address_space(1))) double *base =
        (__attribute__((address_space(1))) double *)(part) +
        partial_index((long)colgroup * (launch_flags & DLAGP_LAUNCH_PIECES_MASK) + split, nblocks, b * ngroups + group, rw, 0, 0);
    typedef v2d __attribute__((address_space(1))) *gstore_d2;
    const gstore_d2 tiles = (gstore_d2)base + lane;
#pragma unroll
    for (int t = 0; t < DLAGP_MFMA_NT; ++t) {
        tiles[(t * 2 + 0) * 64] = v2d{acc[t][0], acc[t][1]};
        tiles[(t * 2 + 1) * 64] = v2d{acc[t][2], acc[t][3]};
    }
    {
        // 4x4x4 outputs: lane = row * 16 + block * 4 + column  ->  sample = block * 4 + row
        const int smp = ((lane >> 2) & 3) * 4 + (lane >> 4), j = lane & 3;
        if (smp < rows_left) {
            if (j < 2) base[DLAGP_NATIVE_EXTRAS + j * 16 + smp] = acc_sw;  // vech columns 208, 209
            base[DLAGP_NATIVE_EXTRAS + (2 + j) * 16 + smp] = acc_sq;       // u 16..19
        }
    }
    double logd = dprod.log_value();
    sum_r2 += __shfl_xor(sum_r2, 16);
    sum_r2 += __shfl_xor(sum_r2, 32);
    logd += __shfl_xor(logd, 16);
    logd += __shfl_xor(logd, 32);
    if (slot == 0 && s_loc < rows_left) {
        base[DLAGP_NATIVE_EXTRAS + 6 * 16 + s_loc] = sum_r2;
        base[DLAGP_NATIVE_EXTRAS + 7 * 16 + s_loc] = logd;
        base[DLAGP_NATIVE_EXTRAS + 8 * 16 + s_loc] = split == 0 ? sl.n * DLAGP_LOG_2PI : 0.0;
    }
    }
#ifdef DLAGP_TRACE_BUILD
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        unsigned long long *t = DLAGP_TRACE + 4 * blockIdx.x;
        t[0] = trace_start;
        t[1] = __builtin_amdgcn_s_memrealtime();
        t[2] = hw_id | ((unsigned long long)xcc_id << 32);
        t[3] = (unsigned long long)(pass_end - pass_begin) | ((unsigned long long)split << 32);
    }
#endif
    if (stamps && lane == 0) {
        const unsigned long long now = stamp_now();
        atomicAdd(&DLAGP_STAMPS[0], t_voigt);
        atomicAdd(&DLAGP_STAMPS[1], t_conv);
        atomicAdd(&DLAGP_STAMPS[2], t_mfma);
        atomicAdd(&DLAGP_STAMPS[3], now - t_mark);
        atomicAdd(&DLAGP_STAMPS[4], now - t_start);
        atomicAdd(&DLAGP_STAMPS[5], 1ULL);
        atomicAdd(&DLAGP_STAMPS[6], t_start - t_entry);  // prologue: descriptor, per-sample constants, core patch table
    }
}

// The kernel: one wave per work item, see accumulate_mfma_body.
template <int LINES, bool BROADEN, bool STAMPS, int MODE = 0, int PATCH = 0, bool LLS = false, bool FUSED = false, bool PAIR = false>
__global__ __launch_bounds__(DLAGP_MFMA_THREADS, 2) void accumulate_mfma(
    const SightlineDev *__restrict__ sightlines, const PassDesc *__restrict__ passes,
    const WorkItem *__restrict__ items, int nb, long S, int kd,
    SampleSource src, double *__restrict__ part, int *__restrict__ status, int launch_flags, double *__restrict__ out,
    double *__restrict__ out2) {
    accumulate_mfma_body<LINES, BROADEN, STAMPS, MODE, PATCH, LLS, FUSED, PAIR>(sightlines, passes, items, nb, S, kd, src, part,
                                                                               status, launch_flags, out, out2);
}

// Diagnostic (dlagp_debug_voigt): the Voigt profile V(v; sigma, gamma_l) = Re w((v + i gamma_l)/(sqrt2 sigma)) /
// (sqrt(2 pi) sigma) of reference voigt.py:241-248 exactly as the kernels form it -- form 0: generic kernels
// (line_optical_depth: asymptotic series / Taylor table), form 1: MFMA kernel (degree-4 wing fit with the
// constants folded per line / Taylor table inside 16 Doppler widths) -- so that tests can hold the device
// Faddeeva evaluation itself to the reference's golden grid.
__global__ void voigt_forms(int form, int line, long n, const double *__restrict__ velocity, double inv_sqrt_2pi_sigma,
                            double *__restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = velocity[i] * DLAGP_LINES.inv_sqrt2_sigma;
    const double x2 = x * x;
    double term;  // lc_l V / lc_l * k_core, i.e. k_core Re w
    if (!(x2 >= DLAGP_FAD_XSPLIT * DLAGP_FAD_XSPLIT)) term = DLAGP_LINES.k_core[line] * core_taylor(x, DLAGP_LINES.y[line]);
    else if (form == 0) term = DLAGP_LINES.k_wing[line] * wing_series(x, DLAGP_LINES.y[line] * DLAGP_LINES.y[line]);
    else term = wing_term(x2, DLAGP_LINES.wing[line]);
    out[i] = term / DLAGP_LINES.k_core[line] * inv_sqrt_2pi_sigma;
}

}  // namespace dlagp
