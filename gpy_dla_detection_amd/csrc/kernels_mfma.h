// MFMA evaluation path for rank k <= 20 (zero-padded to 20): the production kernel.
//
// Work decomposition (DESIGN.md "accumulate_mfma"):
//   workgroup = 4 waves; each wave owns 16 QMC samples of one sightline and one pixel split.
//   The Gram/projection contraction is the FP64 GEMM  [samples x pixels] . [pixels x 256 cols]:
//     A operand (16 samples x 4 pixels per v_mfma_f64_16x16x4_f64) = the per-pixel weights
//       w = a^2/d (14 w-tiles, vech(M_i M_i')) and q = a r/d (2 q-tiles, M_i), generated in
//       registers by the same lane that feeds them to the MFMA -- they never touch memory;
//     B operand = the sightline's shared operand P (kernels_common.h), staged through LDS in
//       chunks of 8 pixels and shared by the 4 waves.
//   Accumulators: 16 tiles x 4 doubles per lane (128 VGPRs) -> 2 waves per SIMD, so one wave's
//   VALU phases (Voigt profile, weights) overlap the sibling wave's MFMAs.
//
// Per pass (<= 56 kept pixels whose 7-tap windows fit in 64 consecutive entries):
//   phase 1  lane = entry, loop over the wave's 16 samples: optical depth of every absorber and
//            line (wing polynomial; Taylor/table core path behind a wave-uniform branch),
//            raw = exp(-N tau) -> LDS raw[sample][entry]            (voigt.py:299-309)
//   phase 2  lane = (sample = lane & 15, pixel slot = lane >> 4), per step of 4 pixels:
//            a = sum_t taps_t raw[pos + t] (voigt.py:319, dla_gp.py:388), d = omega2 a^2 + v,
//            r = y - mu a, w = a^2/d, q = a r/d, running sum r^2/d and product of d
//            (dla_gp.py:392-394, null_gp.py:329-333,356), then 16 MFMAs   (null_gp.py:339-354)
// Output: one partial row per (split, sample) for finalize_samples.
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
#define DLAGP_MFMA_WAVES 4
#define DLAGP_MFMA_THREADS (64 * DLAGP_MFMA_WAVES)
#define DLAGP_MFMA_GROUP (16 * DLAGP_MFMA_WAVES)  // samples per workgroup
#define DLAGP_PASS_ENTRIES 64
#define DLAGP_RAW_STRIDE 66      // doubles per sample row in LDS; stride/2 odd -> conflict-free b64 reads
#define DLAGP_CHUNK_PIXELS 8     // P staging granule = 2 steps
#define DLAGP_CHUNK_DOUBLES (2 * DLAGP_MFMA_NT * DLAGP_TILE_DOUBLES)
#define DLAGP_MAX_PASS_CHUNKS 7  // 56 pixels
#define DLAGP_AUX_FIELDS 8       // per-pixel record: y, mu, omega2, v, window start, 3 spare
#define DLAGP_AUX_DOUBLES (DLAGP_CHUNK_PIXELS * DLAGP_AUX_FIELDS)

struct PassDesc {
    int entry_start;  // first entry computed by phase 1
    int chunk_begin;  // first chunk (8 kept pixels) consumed by phase 2
    int nchunks;      // 1..7
    int pad;
};

// Per-(sample, absorber) constants prepared once per launch: the L velocity multipliers
// c / (lambda_l (1 + z)) / 1e8 (voigt.py:299) followed by the column density.
// Layout: prep[row][absorber][num_lines + 1].
__global__ void prepare_samples(const double *__restrict__ z, const double *__restrict__ nhi, long count,
                                int num_lines, double *__restrict__ prep) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const double zi = z[idx];
    double *dst = prep + idx * (num_lines + 1);
    for (int l = 0; l < num_lines; ++l) dst[l] = line_multiplier(l, zi);
    dst[num_lines] = nhi[idx];
}

// optical depth per unit column density at one entry, all lines; `mult` is wave-uniform
template <int LINES>
__device__ __forceinline__ double optical_depth(double lam, const double *__restrict__ mult, int num_lines) {
    const int L = LINES > 0 ? LINES : num_lines;
    double tau = 0.0;
    double xs[LINES > 0 ? LINES : 1];
    bool any_core = false;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const double velocity = __dsub_rn(__dmul_rn(lam, mult[l]), DLAGP_LINES.c);  // voigt.py:305
        const double x = velocity * DLAGP_LINES.inv_sqrt2_sigma;
        if (LINES > 0) xs[l] = x;
        const double x2 = x * x;
        const bool core = !(x2 >= DLAGP_FAD_XSPLIT * DLAGP_FAD_XSPLIT);  // also catches NaN
        const double u = fast_rcp(fmax(x2, DLAGP_FAD_XSPLIT * DLAGP_FAD_XSPLIT));
        const double *w = DLAGP_LINES.wing[l];
        double poly = w[DLAGP_WING_DEGREE];
#pragma unroll
        for (int m = DLAGP_WING_DEGREE - 1; m >= 0; --m) poly = fma(poly, u, w[m]);
        double term = u * poly;
        if (LINES > 0) {
            any_core |= core;
            tau += core ? 0.0 : term;
        } else {
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

// grid: (ngroups * nsplit, nb); block: 256.
//   block L -> split = L % nsplit (so that blocks sharing an XCD share a slice of P when nsplit
//   is a multiple of 8), sample group = L / nsplit.
// MULTI = false: exactly one absorber per sample, convolution fused into the MFMA loop.
// MULTI = true : kd absorbers; each absorber's profile is convolved and multiplied into
//                per-step registers before the MFMA loop (dla_gp.py:379-386).
// Dynamic LDS: the wave's per-sample constants, 4 waves x 16 samples x kd x (L + 1) doubles.
template <int LINES, bool MULTI>
__global__ __launch_bounds__(DLAGP_MFMA_THREADS, 2) void accumulate_mfma(
    const SightlineDev *__restrict__ sightlines, const PassDesc *const *__restrict__ pass_tables,
    const int2 *const *__restrict__ split_tables, int nsplit, long S, int kd,
    const double *__restrict__ prep, double *__restrict__ part, long rows) {
    __shared__ double raw_lds[DLAGP_MFMA_WAVES][16 * DLAGP_RAW_STRIDE];
    __shared__ double p_lds[2][DLAGP_CHUNK_DOUBLES];
    __shared__ double aux_lds[2][DLAGP_AUX_DOUBLES];
    extern __shared__ double prep_lds[];

    const int b = blockIdx.y;
    const SightlineDev &sl = sightlines[b];
    const int split = blockIdx.x % nsplit;
    const int group = blockIdx.x / nsplit;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s_loc = lane & 15;
    const int slot = lane >> 4;

    // descriptor fields used in the loops, read once
    const int num_entries = sl.num_entries;
    const int ntaps = sl.ntaps;
    const double *__restrict__ lam_ptr = sl.lam;
    const double *__restrict__ p_ptr = sl.P;
    const double *__restrict__ aux_ptr = sl.aux;
    const int *__restrict__ pos_ptr = sl.pos;
    double tap[DLAGP_NUM_TAPS];
#pragma unroll
    for (int t = 0; t < DLAGP_NUM_TAPS; ++t) tap[t] = sl.taps[t];

    const long sample0 = (long)group * DLAGP_MFMA_GROUP + wave * 16;  // first sample of this wave
    const int L = LINES > 0 ? LINES : sl.num_lines;
    const int prep_stride = L + 1;
    double *raw = raw_lds[wave];

    // per-sample constants of this wave -> LDS (rows beyond S repeat the last sample; never stored)
    const int wave_prep = 16 * kd * prep_stride;
    double *my_prep = prep_lds + wave * wave_prep;
    for (int i = lane; i < wave_prep; i += 64) {
        long row = sample0 + i / (kd * prep_stride);
        row = row < S ? row : S - 1;
        my_prep[i] = prep[((long)b * S + row) * kd * prep_stride + i % (kd * prep_stride)];
    }

    const PassDesc *passes = pass_tables[b];
    const int2 span = split_tables[b][split];  // passes [x, y)

    d4 acc[DLAGP_MFMA_NT];
#pragma unroll
    for (int t = 0; t < DLAGP_MFMA_NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    double sum_r2 = 0.0;
    LogProduct dprod;

    // stage the first chunk of this split: P block and the per-pixel record block
    const int first_chunk = span.x < span.y ? passes[span.x].chunk_begin : 0;
    const int end_chunk = span.x < span.y ? passes[span.y - 1].chunk_begin + passes[span.y - 1].nchunks : 0;
    if (first_chunk < end_chunk) {
        const double *src = p_ptr + (long)first_chunk * DLAGP_CHUNK_DOUBLES;
#pragma unroll
        for (int q = 0; q < DLAGP_CHUNK_DOUBLES / DLAGP_MFMA_THREADS / 2; ++q) {
            const int off = (q * DLAGP_MFMA_THREADS + tid) * 2;
            *reinterpret_cast<double2 *>(&p_lds[0][off]) = *reinterpret_cast<const double2 *>(src + off);
        }
        if (tid < DLAGP_AUX_DOUBLES) aux_lds[0][tid] = aux_ptr[(long)first_chunk * DLAGP_AUX_DOUBLES + tid];
    }
    int buf = 0;

    for (int pass = span.x; pass < span.y; ++pass) {
        const PassDesc pd = passes[pass];

        // ---------------- phase 1: raw profiles of the wave's 16 samples ----------------
        int e = pd.entry_start + lane;
        e = e < num_entries ? e : num_entries - 1;
        const double lam = lam_ptr[e];
        if (!MULTI) {
#pragma unroll 2
            for (int ss = 0; ss < 16; ++ss) {
                const double *sp = my_prep + ss * prep_stride;
                const double tau = optical_depth<LINES>(lam, sp, L);
                raw[ss * DLAGP_RAW_STRIDE + lane] = fast_exp(-sp[L] * tau);  // voigt.py:309
            }
        }

        double a_reg[MULTI ? 2 * DLAGP_MAX_PASS_CHUNKS : 1];
        if (MULTI) {
            for (int jd = 0; jd < kd; ++jd) {
#pragma unroll 2
                for (int ss = 0; ss < 16; ++ss) {
                    const double *sp = my_prep + (ss * kd + jd) * prep_stride;
                    const double tau = optical_depth<LINES>(lam, sp, L);
                    raw[ss * DLAGP_RAW_STRIDE + lane] = fast_exp(-sp[L] * tau);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int st = 0; st < 2 * DLAGP_MAX_PASS_CHUNKS; ++st) {
                    if (st < 2 * pd.nchunks) {
                        const int pix = (pd.chunk_begin * 2 + st) * DLAGP_STEP_PIXELS + slot;
                        const double *win = raw + s_loc * DLAGP_RAW_STRIDE + (pos_ptr[pix] - pd.entry_start);
                        const double conv = convolve_window(win, ntaps, tap);
                        a_reg[st] = jd == 0 ? conv : a_reg[st] * conv;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();  // raw[] visible to the whole wave; first chunk staged

        // ---------------- phase 2: weights and MFMAs, chunk by chunk ----------------
        constexpr int CHUNK_UNROLL = MULTI ? DLAGP_MAX_PASS_CHUNKS : 1;
#pragma unroll CHUNK_UNROLL
        for (int c = 0; c < (MULTI ? DLAGP_MAX_PASS_CHUNKS : pd.nchunks); ++c) {
            if (MULTI && c >= pd.nchunks) break;
            const int chunk = pd.chunk_begin + c;
            // prefetch the next chunk (P block + per-pixel records) into registers
            double2 pre[DLAGP_CHUNK_DOUBLES / DLAGP_MFMA_THREADS / 2];
            double pre_aux = 0.0;
            const bool have_next = chunk + 1 < end_chunk;
            if (have_next) {
                const double *src = p_ptr + (long)(chunk + 1) * DLAGP_CHUNK_DOUBLES;
#pragma unroll
                for (int q = 0; q < DLAGP_CHUNK_DOUBLES / DLAGP_MFMA_THREADS / 2; ++q)
                    pre[q] = *reinterpret_cast<const double2 *>(src + (q * DLAGP_MFMA_THREADS + tid) * 2);
                if (tid < DLAGP_AUX_DOUBLES) pre_aux = aux_ptr[(long)(chunk + 1) * DLAGP_AUX_DOUBLES + tid];
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                // per-pixel record: y, mu, omega2, v, window start (as double)
                const double *rec = &aux_lds[buf][(st * DLAGP_STEP_PIXELS + slot) * DLAGP_AUX_FIELDS];
                double a;
                if (MULTI) {
                    a = a_reg[2 * c + st];
                } else {
                    const double *win = raw + s_loc * DLAGP_RAW_STRIDE + ((int)rec[4] - pd.entry_start);
                    a = convolve_window(win, ntaps, tap);
                }
                const double d = fma(rec[2] * a, a, rec[3]);
                const double h = fast_rcp(d);
                const double r = fma(-rec[1], a, rec[0]);
                const double rh = r * h;
                const double w = a * a * h;
                const double q = a * rh;
                sum_r2 = fma(r, rh, sum_r2);
                dprod.multiply(d);
                const double *pb = &p_lds[buf][st * DLAGP_MFMA_NT * DLAGP_TILE_DOUBLES + lane];
#pragma unroll
                for (int t = 0; t < DLAGP_MFMA_NT; ++t) {
                    const double bfrag = pb[t * DLAGP_TILE_DOUBLES];
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(t < DLAGP_MFMA_NTW ? w : q, bfrag, acc[t], 0, 0, 0);
                }
            }
            if (have_next) {
#pragma unroll
                for (int q = 0; q < DLAGP_CHUNK_DOUBLES / DLAGP_MFMA_THREADS / 2; ++q)
                    *reinterpret_cast<double2 *>(&p_lds[buf ^ 1][(q * DLAGP_MFMA_THREADS + tid) * 2]) = pre[q];
                if (tid < DLAGP_AUX_DOUBLES) aux_lds[buf ^ 1][tid] = pre_aux;
            }
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---------------- epilogue: partial sums to part[split][column][row] ----------------
    // Transposed through LDS (p_lds is free now) so that every store instruction writes the 64
    // samples of one column contiguously.  D layout: lane holds column (lane & 15) of rows
    // (lane >> 4) + 4 r of its wave's 16-sample tile.
    const int rw = sl.row_width;
    const long row_base = (long)b * S + (long)group * DLAGP_MFMA_GROUP;  // first row of this workgroup
    const long rows_left = S - (long)group * DLAGP_MFMA_GROUP;            // valid samples from there
    double *base = part + (long)split * rw * rows;
    double *tr = &p_lds[0][0];           // [16 columns][65] doubles per tile
    constexpr int TR_STRIDE = 65;
#pragma unroll
    for (int t = 0; t < DLAGP_MFMA_NT; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) tr[s_loc * TR_STRIDE + wave * 16 + slot + 4 * r] = acc[t][r];
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < DLAGP_TILE_COLS / DLAGP_MFMA_WAVES; ++cc) {
            const int col = wave * (DLAGP_TILE_COLS / DLAGP_MFMA_WAVES) + cc;
            if (lane < rows_left) base[(long)(t * DLAGP_TILE_COLS + col) * rows + row_base + lane] = tr[col * TR_STRIDE + lane];
        }
    }
    double logd = dprod.log_value();
    sum_r2 += __shfl_xor(sum_r2, 16);
    sum_r2 += __shfl_xor(sum_r2, 32);
    logd += __shfl_xor(logd, 16);
    logd += __shfl_xor(logd, 32);
    if (slot == 0 && wave * 16 + s_loc < rows_left) {
        const long row = row_base + wave * 16 + s_loc;
        base[(long)(DLAGP_MFMA_NT * DLAGP_TILE_COLS) * rows + row] = sum_r2;
        base[(long)(DLAGP_MFMA_NT * DLAGP_TILE_COLS + 1) * rows + row] = logd;
    }
}

}  // namespace dlagp
