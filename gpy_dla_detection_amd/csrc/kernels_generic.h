// Generic evaluation path: one 256-thread workgroup per QMC sample.
//
// Handles every rank k <= DLAGP_MAX_RANK, any number of Lyman lines and absorbers.  It is the
// path for shapes the MFMA kernel does not specialise and an independent cross-check of it;
// its structure follows the reference one sample at a time:
//   phase A  raw profile of every absorber on the entries, instrument convolution, product
//            (voigt.py:299-319, dla_gp.py:370-388)                      -> a[i] in LDS
//   phase B  d = omega2 a^2 + v, r = y - mu a, w = a^2/d, q = a r/d,
//            sum r^2/d, sum log d (dla_gp.py:392-394, null_gp.py:329-333,356)
//   phase C  one thread per column of P: sum_i w_i P_i (B - I) and sum_i q_i M_i (u)
// and writes one partial column set [B-I | u | sum r^2/d, sum log d] for finalize_samples
// (layout partial_index, see kernels_common.h).
#pragma once
#include <hip/hip_runtime.h>

#include "dlagp_device.h"
#include "kernels_common.h"
#include "voigt_device.h"

namespace dlagp {


// Absorption vector of one sample into LDS a[0..n_pad) (padding pixels get 1).
// raw[] is scratch of num_entries doubles; mult[] of DLAGP_MAX_LINES doubles.
__device__ __forceinline__ void sample_absorption(const SightlineDev &sl, int kd, const SampleSource &src, long b,
                                                  long smp, long S, double *raw, double *a, double *mult) {
    const int tid = threadIdx.x;
    if (kd == 0) {  // null model: a = 1 exactly
        for (int i = tid; i < sl.n_pad; i += DLAGP_GENERIC_THREADS) a[i] = 1.0;
        __syncthreads();
        return;
    }
    for (int j = 0; j < kd; ++j) {
        const double zj = sample_z(src, b, smp, j, S, kd);
        const double nj = sample_nhi(src, b, smp, j, S, kd);
        if (tid < sl.num_lines) mult[tid] = line_multiplier(tid, zj);
        __syncthreads();
        for (int e = tid; e < sl.num_entries; e += DLAGP_GENERIC_THREADS) {
            const double lam = sl.lam[e];
            double exponent = -nj * line_optical_depth(lam, mult, sl.num_lines);  // voigt.py:309
            if (sl.lls_break) {
                // Lyman-limit break, voigt_lls.py:271-284,348: N / 10^17.2 (lambda_rest / 911.7641)^3 bluewards of the limit
                const double ratio = lam / (1.0 + zj) / DLAGP_LYMAN_LIMIT_ANGSTROM;
                if (!(ratio > 1.0)) exponent -= nj / DLAGP_LLS_COLUMN_SCALE * (ratio * ratio * ratio);
            }
            raw[e] = exp(exponent);
        }
        __syncthreads();
        for (int i = tid; i < sl.n_pad; i += DLAGP_GENERIC_THREADS) {
            double conv = 1.0;
            if (i < sl.n) {
                const int p = sl.pos[i];
                conv = 0.0;
                for (int t = 0; t < sl.ntaps; ++t) conv = fma(raw[p + t], sl.taps[t], conv);  // voigt.py:319
            }
            a[i] = j == 0 ? conv : a[i] * conv;  // dla_gp.py:379-386
        }
        __syncthreads();
    }
}

__device__ __forceinline__ double block_sum(double x, double *red) {
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = x;
    __syncthreads();
    double total = 0.0;
    for (int w = 0; w < DLAGP_GENERIC_THREADS / 64; ++w) total += red[w];
    __syncthreads();
    return total;
}

// grid: (S, nb, pixel splits).  Dynamic LDS: (num_entries + n_pad + MAX_LINES + 8) doubles (max over sightlines).
// gridDim.z > 1 (the null model: one sample per sightline, nothing to amortise the pixel loop over) cuts every
// sightline's steps into gridDim.z equal ranges, one workgroup and one set of partial sums each (the finalisation adds
// them in order, like the MFMA path's pieces): the null evidence of one spectrum took 68 us on one workgroup.
__global__ __launch_bounds__(DLAGP_GENERIC_THREADS) void accumulate_generic(
    const SightlineDev *__restrict__ sightlines, long S, int kd, SampleSource src, double *__restrict__ part,
    int *__restrict__ status, int lds_entries, int lds_pixels, int keep_status) {
    extern __shared__ double lds[];
    double *raw = lds;
    double *a = raw + lds_entries;
    double *mult = a + lds_pixels;
    double *red = mult + DLAGP_MAX_LINES + 1;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && !keep_status) *status = 0;  // finalize raises it

    const SightlineDev &sl = sightlines[blockIdx.y];
    const int tid = threadIdx.x;

    sample_absorption(sl, kd, src, sl.batch_index, blockIdx.x, S, raw, a, mult);

    // this workgroup's range of steps (all of them unless the launch splits the pixels)
    const int nsplit = gridDim.z, split = blockIdx.z;
    const int steps_per = (sl.nsteps + nsplit - 1) / nsplit;
    const int step_lo = min(split * steps_per, sl.nsteps), step_hi = min(step_lo + steps_per, sl.nsteps);

    // phase B: per-pixel weights; w overwrites raw[], q overwrites a[]
    double *w = raw;
    double sum_r2 = 0.0, sum_logd = 0.0;
    for (int i = step_lo * DLAGP_STEP_PIXELS + tid; i < step_hi * DLAGP_STEP_PIXELS; i += DLAGP_GENERIC_THREADS) {
        const double ai = a[i];
        const double d = fma(sl.omega2[i] * ai, ai, sl.v[i]);
        const double h = 1.0 / d;
        const double r = fma(-sl.mu[i], ai, sl.y[i]);
        const double rh = r * h;
        w[i] = ai * ai * h;
        a[i] = ai * rh;
        sum_r2 = fma(r, rh, sum_r2);
        sum_logd += log(d);
    }
    __syncthreads();
    sum_r2 = block_sum(sum_r2, red);
    sum_logd = block_sum(sum_logd, red);

    // phase C: columns
    const int nt = sl.ntw + sl.ntq;
    const long ngroups = (S + 15) / 16;
    const long block = blockIdx.y * ngroups + blockIdx.x / 16;
    const long nblocks = gridDim.y * ngroups;
    const int lane16 = blockIdx.x % 16;
    for (int c = tid; c < nt * DLAGP_TILE_COLS; c += DLAGP_GENERIC_THREADS) {
        const int tile = c / DLAGP_TILE_COLS;
        const int cc = c % DLAGP_TILE_COLS;
        const double *weight = tile < sl.ntw ? w : a;
        double acc = 0.0;
        for (int step = step_lo; step < step_hi; ++step) {
#pragma unroll
            for (int slot = 0; slot < DLAGP_STEP_PIXELS; ++slot)
                acc = fma(weight[step * DLAGP_STEP_PIXELS + slot],
                          sl.P[shared_operand_index(step, tile, slot, cc, sl.npair)], acc);
        }
        part[partial_index(split, nblocks, block, sl.row_width, c, lane16)] = acc;
    }
    if (tid == 0) {
        part[partial_index(split, nblocks, block, sl.row_width, nt * DLAGP_TILE_COLS, lane16)] = sum_r2;
        part[partial_index(split, nblocks, block, sl.row_width, nt * DLAGP_TILE_COLS + 1, lane16)] = sum_logd;
        part[partial_index(split, nblocks, block, sl.row_width, nt * DLAGP_TILE_COLS + 2, lane16)] = split == 0 ? sl.n * DLAGP_LOG_2PI : 0.0;
    }
}

// Absorption vector of one sample to global memory (dlagp_absorption).  grid: 1 block.
__global__ __launch_bounds__(DLAGP_GENERIC_THREADS) void absorption_generic(
    const SightlineDev *__restrict__ sightlines, int kd, const double *__restrict__ z,
    const double *__restrict__ nhi, double *__restrict__ out, int lds_entries, int lds_pixels) {
    extern __shared__ double lds[];
    double *raw = lds;
    double *a = raw + lds_entries;
    double *mult = a + lds_pixels;
    const SightlineDev &sl = sightlines[0];
    const SampleSource src{z, nhi, nullptr, 0, 0, 0, nullptr, nullptr, 0, 0, 0, 0, 0};
    sample_absorption(sl, kd, src, 0, 0, 1, raw, a, mult);
    for (int i = threadIdx.x; i < sl.n; i += DLAGP_GENERIC_THREADS) out[i] = a[i];
}

}  // namespace dlagp
