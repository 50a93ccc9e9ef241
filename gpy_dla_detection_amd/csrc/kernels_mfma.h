// MFMA evaluation path for rank k <= 20 (zero-padded to 20): the production kernel.
//
// On gfx950 the FP64 matrix and FP64 vector instructions share one execution pipe
// (profiles/r01_microbench_dp_pipe.txt: an MFMA wave and an FMA wave on one SIMD take the SUM
// of their times), so there is no "free" VALU time under the MFMAs.  The kernel is therefore
// built to (a) spend as few VALU instructions per (sample, pixel) as possible and (b) never
// stall: every wave is independent -- no workgroup barriers, no shared staging -- and streams
// its operands from L2 one step ahead of use.
//
// Work decomposition (DESIGN.md "accumulate_mfma"):
//   one wave (= one 64-thread workgroup) owns 16 QMC samples of one sightline and one pixel split.
//   The Gram/projection contraction is the FP64 GEMM  [samples x pixels] . [pixels x 256 cols]:
//     A operand (16 samples x 4 pixels per v_mfma_f64_16x16x4_f64) = the per-pixel weights
//       w = a^2/d (14 w-tiles, vech(M_i M_i')) and q = a r/d (2 q-tiles, M_i), generated in
//       registers by the same lane that feeds them to the MFMA -- they never touch memory;
//     B operand = the sightline's shared operand P (kernels_common.h), read straight from L2 into
//       registers as 8 double2 per lane and step; each register pair is reloaded for the next
//       step right after its two MFMAs have issued (a rolling one-step-ahead prefetch).
//   Accumulators: 16 tiles x 4 doubles per lane (128 VGPRs) -> 2 waves per SIMD.
//
// Per pass (<= 56 kept pixels whose 7-tap windows fit in 64 consecutive entries):
//   phase 1  lane = entry, loop over the wave's 16 samples: optical depth of every absorber and
//            line (wing polynomial; Taylor/table core path behind a wave-uniform branch),
//            raw = exp(-N tau) -> LDS raw[sample][entry]            (voigt.py:299-309)
//   phase 2  lane = (sample = lane & 15, pixel slot = lane >> 4), per step of 4 pixels:
//            a = sum_t taps_t raw[pos + t] (voigt.py:319, dla_gp.py:388), d = omega2 a^2 + v,
//            r = y - mu a, w = a^2/d, q = a r/d, running sum r^2/d and product of d
//            (dla_gp.py:392-394, null_gp.py:329-333,356), then 16 MFMAs   (null_gp.py:339-354)
// Output: partial sums part[split][column][row] for the finalize kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "dlagp_device.h"
#include "fastmath_device.h"
#include "voigt_device.h"

namespace dlagp {

typedef double d4 __attribute__((ext_vector_type(4)));

#define DLAGP_MFMA_RANK 20
#define DLAGP_MFMA_NTW 14
#define DLAGP_MFMA_NT 16
#define DLAGP_MFMA_GROUP 16      // samples per wave / workgroup
#define DLAGP_MFMA_THREADS 64
#define DLAGP_PASS_ENTRIES 64
#define DLAGP_RAW_STRIDE 66      // doubles per sample row in LDS; stride/2 odd -> conflict-free b64 reads
#define DLAGP_CHUNK_PIXELS 8     // planning granule = 2 steps
#define DLAGP_MAX_PASS_CHUNKS 7  // 56 pixels
#define DLAGP_MAX_PASS_STEPS (2 * DLAGP_MAX_PASS_CHUNKS)
#define DLAGP_AUX_FIELDS 8       // per-pixel record: y, mu, omega2, v, window start, 3 spare

struct PassDesc {
    int entry_start;  // first entry computed by phase 1
    int chunk_begin;  // first chunk (8 kept pixels) consumed by phase 2
    int nchunks;      // 1..7
    int pad;
};

// Per-(sample, absorber) constants prepared once per launch:
//   m'_l = c / (lambda_l (1 + z)) / 1e8 / (sqrt2 sigma)   so that  x_l = lam m'_l - c / (sqrt2 sigma)
// (voigt.py:299,305,247 folded into one FMA per line), followed by the column density.
// Layout: prep[row][absorber][num_lines + 1].
__global__ void prepare_samples(const double *__restrict__ z, const double *__restrict__ nhi, long count,
                                int num_lines, double *__restrict__ prep) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const double zi = z[idx];
    double *dst = prep + idx * (num_lines + 1);
    for (int l = 0; l < num_lines; ++l) dst[l] = line_multiplier(l, zi) * DLAGP_LINES.inv_sqrt2_sigma;
    dst[num_lines] = nhi[idx];
}

// optical depth per unit column density at one entry, all lines; `mult` is wave-uniform (LDS)
template <int LINES>
__device__ __forceinline__ double optical_depth(double lam, const double *__restrict__ mult, int num_lines) {
    const int L = LINES > 0 ? LINES : num_lines;
    const double x_offset = -DLAGP_LINES.c * DLAGP_LINES.inv_sqrt2_sigma;
    double tau = 0.0;
    double xs[LINES > 0 ? LINES : 1];
    bool any_core = false;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const double x = fma(lam, mult[l], x_offset);
        if (LINES > 0) xs[l] = x;
        const double x2 = x * x;
        const bool core = !(x2 >= DLAGP_FAD_XSPLIT * DLAGP_FAD_XSPLIT);  // also catches NaN
        // core lanes get a finite dummy argument; their term is discarded below
        const double u = fast_rcp1(core ? DLAGP_FAD_XSPLIT * DLAGP_FAD_XSPLIT : x2);
        const double *w = DLAGP_LINES.wing[l];
        double poly = w[DLAGP_WING_DEGREE];
#pragma unroll
        for (int m = DLAGP_WING_DEGREE - 1; m >= 0; --m) poly = fma(poly, u, w[m]);
        if (LINES > 0) {
            any_core |= core;
            tau = fma(core ? 0.0 : u, poly, tau);
        } else {
            double term = u * poly;
            if (core) term = DLAGP_LINES.k_core[l] * core_taylor(x, DLAGP_LINES.y[l]);
            if (term == term) tau += term;  // np.nansum (voigt.py:309)
        }
    }
    if (LINES > 0) {
        // rare: some lane sits within 16 Doppler widths of a line centre (~6 entries per line)
        if (__builtin_amdgcn_ballot_w64(any_core)) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const double x = xs[l];
                if (!(x * x >= DLAGP_FAD_XSPLIT * DLAGP_FAD_XSPLIT)) {
                    const double term = DLAGP_LINES.k_core[l] * core_taylor(x, DLAGP_LINES.y[l]);
                    if (term == term) tau += term;
                }
            }
        }
    }
    return tau;
}

// 7-tap (or 1-tap) window of the raw profile = instrument convolution (voigt.py:319)
__device__ __forceinline__ double convolve_window(const double *win, int ntaps, const double (&tap)[DLAGP_NUM_TAPS]) {
    double conv = win[0] * tap[0];
    if (ntaps > 1) {
#pragma unroll
        for (int t = 1; t < DLAGP_NUM_TAPS; ++t) conv = fma(win[t], tap[t], conv);
    }
    return conv;
}

// grid: (ngroups * nsplit, nb); block: one wave.
//   block L -> split = L % nsplit, sample group (16 samples) = L / nsplit.
// MULTI = false: exactly one absorber per sample, convolution fused into the MFMA loop.
// MULTI = true : kd absorbers; each absorber's profile is convolved and multiplied into an LDS
//                buffer a[sample][pixel of the pass] before the MFMA loop (dla_gp.py:379-386).
// Dynamic LDS: the wave's per-sample constants, 16 samples x kd x (L + 1) doubles.
template <int LINES, bool MULTI>
__global__ __launch_bounds__(DLAGP_MFMA_THREADS, 2) void accumulate_mfma(
    const SightlineDev *__restrict__ sightlines, const PassDesc *const *__restrict__ pass_tables,
    const int2 *const *__restrict__ split_tables, int nsplit, long S, int kd,
    const double *__restrict__ prep, double *__restrict__ part, long rows) {
    __shared__ double raw[16 * DLAGP_RAW_STRIDE];
    __shared__ double a_lds[MULTI ? 16 * (4 * DLAGP_MAX_PASS_STEPS + 2) : 2];
    extern __shared__ double prep_lds[];
    constexpr int A_STRIDE = 4 * DLAGP_MAX_PASS_STEPS + 2;

    const int b = blockIdx.y;
    const SightlineDev &sl = sightlines[b];
    const int split = blockIdx.x % nsplit;
    const long group = blockIdx.x / nsplit;
    const int lane = threadIdx.x;
    const int s_loc = lane & 15;
    const int slot = lane >> 4;

    // descriptor fields used in the loops, read once
    const int num_entries = sl.num_entries;
    const int ntaps = sl.ntaps;
    const double *__restrict__ lam_ptr = sl.lam;
    const double2 *__restrict__ p_ptr = reinterpret_cast<const double2 *>(sl.P);
    const double *__restrict__ aux_ptr = sl.aux;
    double tap[DLAGP_NUM_TAPS];
#pragma unroll
    for (int t = 0; t < DLAGP_NUM_TAPS; ++t) tap[t] = sl.taps[t];

    const long sample0 = group * DLAGP_MFMA_GROUP;  // first sample of this wave
    const int L = LINES > 0 ? LINES : sl.num_lines;
    const int prep_stride = L + 1;

    // per-sample constants of this wave -> LDS (rows beyond S repeat the last sample; never stored)
    const int wave_prep = 16 * kd * prep_stride;
    for (int i = lane; i < wave_prep; i += 64) {
        long row = sample0 + i / (kd * prep_stride);
        row = row < S ? row : S - 1;
        prep_lds[i] = prep[((long)b * S + row) * kd * prep_stride + i % (kd * prep_stride)];
    }

    const PassDesc *passes = pass_tables[b];
    const int2 span = split_tables[b][split];  // passes [x, y)
    const int first_step = span.x < span.y ? 2 * passes[span.x].chunk_begin : 0;
    const int end_step = span.x < span.y ? 2 * (passes[span.y - 1].chunk_begin + passes[span.y - 1].nchunks) : 0;

    d4 acc[DLAGP_MFMA_NT];
#pragma unroll
    for (int t = 0; t < DLAGP_MFMA_NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    double sum_r2 = 0.0;
    LogProduct dprod;

    // B operand of the first step: 8 x double2 per lane (tile pair j -> tiles 2j, 2j+1)
    constexpr int NPAIR = DLAGP_MFMA_NT / 2;
    double2 frag[NPAIR];
    {
        const double2 *src = p_ptr + (long)first_step * NPAIR * 64 + lane;
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) frag[j] = src[j * 64];
    }
    // per-pixel record of the first step: y, mu, omega2, v, window start
    const double *rec_src = aux_ptr + ((long)first_step * DLAGP_STEP_PIXELS + slot) * DLAGP_AUX_FIELDS;
    double rec_y = rec_src[0], rec_mu = rec_src[1], rec_o = rec_src[2], rec_v = rec_src[3], rec_pos = rec_src[4];

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    for (int pass = span.x; pass < span.y; ++pass) {
        const PassDesc pd = passes[pass];
        const int pass_steps = 2 * pd.nchunks;
        const int step0 = 2 * pd.chunk_begin;

        // ---------------- phase 1: raw profiles of the wave's 16 samples ----------------
        int e = pd.entry_start + lane;
        e = e < num_entries ? e : num_entries - 1;
        const double lam = lam_ptr[e];
        if (!MULTI) {
#pragma unroll 2
            for (int ss = 0; ss < 16; ++ss) {
                const double *sp = prep_lds + ss * prep_stride;
                const double tau = optical_depth<LINES>(lam, sp, L);
                raw[ss * DLAGP_RAW_STRIDE + lane] = fast_exp(-sp[L] * tau);  // voigt.py:309
            }
        } else {
            for (int jd = 0; jd < kd; ++jd) {
#pragma unroll 2
                for (int ss = 0; ss < 16; ++ss) {
                    const double *sp = prep_lds + (ss * kd + jd) * prep_stride;
                    const double tau = optical_depth<LINES>(lam, sp, L);
                    raw[ss * DLAGP_RAW_STRIDE + lane] = fast_exp(-sp[L] * tau);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                for (int st = 0; st < pass_steps; ++st) {
                    const int pix = (step0 + st) * DLAGP_STEP_PIXELS + slot;
                    const int p = (int)aux_ptr[(long)pix * DLAGP_AUX_FIELDS + 4] - pd.entry_start;
                    const double conv = convolve_window(raw + s_loc * DLAGP_RAW_STRIDE + p, ntaps, tap);
                    double *dst = &a_lds[s_loc * A_STRIDE + st * DLAGP_STEP_PIXELS + slot];
                    *dst = jd == 0 ? conv : *dst * conv;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        // ---------------- phase 2: weights and MFMAs, step by step ----------------
        for (int st = 0; st < pass_steps; ++st) {
            const int step = step0 + st;
            double a;
            if (MULTI) {
                a = a_lds[s_loc * A_STRIDE + st * DLAGP_STEP_PIXELS + slot];
            } else {
                a = convolve_window(raw + s_loc * DLAGP_RAW_STRIDE + ((int)rec_pos - pd.entry_start), ntaps, tap);
            }
            const double a2 = a * a;
            const double d = fma(rec_o, a2, rec_v);
            const double h = fast_rcp1(d);
            const double r = fma(-rec_mu, a, rec_y);
            const double rh = r * h;
            const double w = a2 * h;
            const double q = a * rh;
            sum_r2 = fma(r, rh, sum_r2);
            dprod.multiply(d);

            // next step's operands (clamped at the end of the split: a harmless re-read)
            const int nxt = step + 1 < end_step ? step + 1 : step;
            const double *nrec = aux_ptr + ((long)nxt * DLAGP_STEP_PIXELS + slot) * DLAGP_AUX_FIELDS;
            rec_y = nrec[0];
            rec_mu = nrec[1];
            rec_o = nrec[2];
            rec_v = nrec[3];
            rec_pos = nrec[4];
            const double2 *src = p_ptr + (long)nxt * NPAIR * 64 + lane;
#pragma unroll
            for (int j = 0; j < NPAIR; ++j) {
                const double2 f = frag[j];
                acc[2 * j] = __builtin_amdgcn_mfma_f64_16x16x4f64(2 * j < DLAGP_MFMA_NTW ? w : q, f.x, acc[2 * j], 0, 0, 0);
                acc[2 * j + 1] =
                    __builtin_amdgcn_mfma_f64_16x16x4f64(2 * j + 1 < DLAGP_MFMA_NTW ? w : q, f.y, acc[2 * j + 1], 0, 0, 0);
                frag[j] = src[j * 64];  // rolling prefetch: this pair is next used one full step later
            }
        }
        // the next pass overwrites raw[] / a_lds[]: order the LDS reads above before those writes
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }

    // ---------------- epilogue: partial sums to part[split][column][row] ----------------
    // D layout: lane holds column (lane & 15) of rows (lane >> 4) + 4 r of the 16-sample tile.
    // Transposed through LDS (raw[] is free now) so that each store writes 16 consecutive samples
    // of 4 columns (4 x 128 B segments).
    const int rw = sl.row_width;
    const long row_base = (long)b * S + sample0;
    const long rows_left = S - sample0;  // valid samples of this wave
    double *base = part + (long)split * rw * rows;
    constexpr int TR_STRIDE = 17;  // [16 columns][16 samples + 1]
#pragma unroll
    for (int t = 0; t < DLAGP_MFMA_NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) raw[s_loc * TR_STRIDE + slot + 4 * r] = acc[t][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int col = cc * 4 + slot;  // lane -> (sample = s_loc, column = col)
            const double val = raw[col * TR_STRIDE + s_loc];
            if (s_loc < rows_left) base[(long)(t * DLAGP_TILE_COLS + col) * rows + row_base + s_loc] = val;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    double logd = dprod.log_value();
    sum_r2 += __shfl_xor(sum_r2, 16);
    sum_r2 += __shfl_xor(sum_r2, 32);
    logd += __shfl_xor(logd, 16);
    logd += __shfl_xor(logd, 32);
    if (slot == 0 && s_loc < rows_left) {
        const long row = row_base + s_loc;
        base[(long)(DLAGP_MFMA_NT * DLAGP_TILE_COLS) * rows + row] = sum_r2;
        base[(long)(DLAGP_MFMA_NT * DLAGP_TILE_COLS + 1) * rows + row] = logd;
    }
}

}  // namespace dlagp
