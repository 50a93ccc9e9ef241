// Kernels shared by every evaluation path: the shared-operand builder and the per-sample
// finalisation (k x k Cholesky, triangular solve, log-determinant, assembly).
#pragma once
#include "host_shared.h"
#include <hip/hip_runtime.h>

#include <type_traits>

#include "dlagp_device.h"
#include "fastmath_device.h"

namespace dlagp {

#define DLAGP_LOG_2PI 1.83787706640934534  // reference null_gp.py:325
#define DLAGP_MAX_ABSORBERS_DEV 8           // = DLAGP_MAX_ABSORBERS of include/dlagp.h
#define DLAGP_LYMAN_LIMIT_ANGSTROM 911.7641             // reference voigt_lls.py:227
#define DLAGP_LLS_COLUMN_SCALE 1.5848931924611142e17    // 10**17.2, reference voigt_lls.py:277

// Partial sums live in blocks of 16 consecutive samples of one sightline:
//   part[split][block][column][16],  block = sightline * ngroups + sample / 16,  ngroups = ceil(S / 16)
// so that one accumulate wave writes, and one finalize workgroup reads, a single contiguous block.
__host__ __device__ __forceinline__ long partial_index(long split, long nblocks, long block, int row_width, int col,
                                                       int lane16) {
    return (((split * nblocks + block) * row_width) + col) * 16 + lane16;
}

// column of B(i, j), j <= i, inside the w-tiles
__host__ __device__ __forceinline__ int vech_col(int i, int j) { return i * (i + 1) / 2 + j; }

// ---------------------------------------------------------------------------------------------
// Shared operand P of one sightline, built once at dlagp_create:
//   w-tiles: P[i][vech_col(a, b)] = M[i][a] M[i][b]   -> B - I = sum_i w_i P_i   (null_gp.py:339)
//   q-tiles: P[i][ntw*16 + c]     = M[i][c]           -> u     = sum_i q_i M_i   (null_gp.py:346-354)
// Padding pixels and padding columns are zero, so they contribute exactly nothing.
// Layout [step][tile pair j][lane][2]: lane = slot * 16 + column-in-tile is the MFMA B-operand
// lane of pixel 4 step + slot, and the two doubles are tiles 2j and 2j + 1, so one 16-byte load
// per lane fetches the operands of two MFMAs.  M_dev is dense row-major [n][k].
__host__ __device__ __forceinline__ long shared_operand_index(int step, int tile, int slot, int cc, int npair) {
    return (((long)step * npair + tile / 2) * DLAGP_TILE_DOUBLES + slot * DLAGP_TILE_COLS + cc) * 2 + (tile & 1);
}

__global__ void build_shared_operand(const double *__restrict__ M_dev, int n, int n_pad, int k, int ntw,
                                     int ntq, int npair, double *__restrict__ P) {
    const int nt = 2 * npair;
    const long total = (long)n_pad * nt * DLAGP_TILE_COLS;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int cc = idx % DLAGP_TILE_COLS;
        const int tile = (idx / DLAGP_TILE_COLS) % nt;
        const int i = idx / ((long)DLAGP_TILE_COLS * nt);
        const int step = i / DLAGP_STEP_PIXELS, slot = i % DLAGP_STEP_PIXELS;
        double val = 0.0;
        if (i < n) {
            if (tile < ntw) {
                const int col = tile * DLAGP_TILE_COLS + cc;
                // invert vech_col: largest a with a (a + 1) / 2 <= col
                int a = (int)((sqrt(8.0 * col + 1.0) - 1.0) * 0.5);
                while ((a + 1) * (a + 2) / 2 <= col) ++a;
                while (a * (a + 1) / 2 > col) --a;
                const int b = col - a * (a + 1) / 2;
                if (a < k) val = M_dev[(long)i * k + a] * M_dev[(long)i * k + b];
            } else if (tile < ntw + ntq) {
                const int c = (tile - ntw) * DLAGP_TILE_COLS + cc;
                if (c < k) val = M_dev[(long)i * k + c];
            }
        }
        P[shared_operand_index(step, tile, slot, cc, npair)] = val;
    }
}

// ---------------------------------------------------------------------------------------------
// Model interpolation: NullGP.get_interp (reference null_gp.py:179-242) on the device.
// Per kept pixel i (32 threads each, see the kernel):
//   linear interpolation of mu, log_omega and the k columns of M at the pixel's rest wavelength
//     (scipy interp1d(kind="linear") = numpy.interp: slope = (f[j+1]-f[j])/(x[j+1]-x[j]),
//      value = slope (x - x[j]) + f[j], null_gp.py:67-71,189-195) -- same operation order, no FMA;
//   Lyman-series effective optical depth, effective_optical_depth.py:43-80:
//     tau_l = scale_l (1 + z_l)^beta [z_l <= z_qso],  z_l = (lambda_obs - lambda_l)/lambda_l,
//     summed over the series in numpy's order for a 31-element row (8 strided partial sums combined
//     pairwise, then the tail);
//   mean-flux suppression exp(-sum tau) on mu and M (null_gp.py:197-214), noise scaling
//     omega2 = exp(2 log_omega) (1 - exp(-sum tau_noise) + c_0)^2 suppression^2 (:216-236).
// Outputs in the layout dlagp_create builds from host arrays: mu, omega2 padded to n_pad with zeros,
// M dense row-major [n][k_out] (columns >= k zero).
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ double numpy_row_sum(const double *t, int n) {
    if (n < 8) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s = __dadd_rn(s, t[i]);
        return s;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = t[j];
    int i = 8;
    for (; i + 8 <= n; i += 8)
        for (int j = 0; j < 8; ++j) r[j] = __dadd_rn(r[j], t[i + j]);
    double s = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                         __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
    for (; i < n; ++i) s = __dadd_rn(s, t[i]);
    return s;
}

__device__ __forceinline__ double numpy_interp(const double *__restrict__ xp, const double *__restrict__ fp, int j,
                                               int n_grid, double x) {
    if (j >= n_grid - 1) return fp[n_grid - 1];
    const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    return __dadd_rn(__dmul_rn(slope, x - xp[j]), fp[j]);
}

// 32 threads per kept pixel (8 pixels per workgroup): thread l evaluates series member l's two optical depths (two
// `pow` calls: one thread per pixel doing all 62 of them made this a 127 us kernel, the longest step of context
// creation), thread 0 sums them in numpy's order and forms the two factors, then the 32 threads share the pixel's
// k + 2 interpolations.  Arithmetic per value unchanged.
static_assert(DLAGP_MAX_LINES <= 32, "one thread per series member");

__global__ __launch_bounds__(DLAGP_INTERP_THREADS) void interp_model(
    ModelDev model, int n, int n_pad, int k_out, const double *__restrict__ x, const double *__restrict__ wavelengths,
    double z_qso, double *__restrict__ mu_out, double *__restrict__ omega2_out, double *__restrict__ M_out) {
    __shared__ double tau_prev[DLAGP_INTERP_PIXELS][32], tau_noise[DLAGP_INTERP_PIXELS][32];
    __shared__ double factor[DLAGP_INTERP_PIXELS][2];
    __shared__ int cell[DLAGP_INTERP_PIXELS];
    const int p = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int i = blockIdx.x * DLAGP_INTERP_PIXELS + p;
    const bool real = i < n;
    const int L = model.num_forest_lines;
    const double xi = real ? x[i] : 0.0;
    if (real && l < L) {
        const double w = wavelengths[i];
        const double z_abs = (w - model.lines[l]) / model.lines[l];
        const double base = 1.0 + z_abs;
        const double inside = z_abs <= z_qso ? 1.0 : 0.0;
        tau_prev[p][l] = model.scale_prev[l] * pow(base, model.prev_beta) * inside;
        tau_noise[p][l] = model.scale_noise[l] * pow(base, model.beta) * inside;
    }
    if (real && l == 31) {
        // largest j with grid[j] <= xi (the host has checked grid[0] <= xi <= grid[n_grid - 1])
        int lo = 0, hi = model.n_grid - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (model.grid[mid] <= xi) lo = mid;
            else hi = mid - 1;
        }
        cell[p] = lo;
    }
    __syncthreads();
    if (real && l == 0) {
        factor[p][0] = exp(-numpy_row_sum(tau_prev[p], L));
        factor[p][1] = __dadd_rn(1.0 - exp(-numpy_row_sum(tau_noise[p], L)), model.c_0);
    }
    __syncthreads();
    if (i >= n_pad) return;
    if (!real) {  // padding pixels
        if (l == 0) {
            mu_out[i] = 0.0;
            omega2_out[i] = 0.0;
        }
        return;
    }
    const int j = cell[p];
    const double absorption = factor[p][0], scaling = factor[p][1];
    for (int c = l; c < k_out + 2; c += 32) {
        if (c < k_out) {
            M_out[(long)i * k_out + c] =
                c < model.k ? numpy_interp(model.grid, model.M + (long)c * model.n_grid, j, model.n_grid, xi) * absorption
                            : 0.0;
        } else if (c == k_out) {
            mu_out[i] = numpy_interp(model.grid, model.mu, j, model.n_grid, xi) * absorption;
        } else {
            double o2 = exp(2.0 * numpy_interp(model.grid, model.log_omega, j, model.n_grid, xi));
            o2 = __dmul_rn(o2, __dmul_rn(scaling, scaling));
            omega2_out[i] = __dmul_rn(o2, __dmul_rn(absorption, absorption));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA operand of one sightline (rank padded to 20), slot-ordered (dlagp_device.h "slot"):
//   Pm[step][pair j][lane][2],  step = 4 consecutive slots.
//   pairs 0..6 hold the 14 full tiles t = 2j, 2j+1 for v_mfma_f64_16x16x4_f64 (B operand: lane = slot-in-step * 16 +
//   column-in-tile):  t < 13: vech(M_i M_i') columns 16t .. 16t+15;  t = 13: M_i columns 0..15.
//   pair 7 holds the operands of two v_mfma_f64_4x4x4_4b_f64 (profiles/r02_microbench_mfma_f64_4x4.txt: B operand
//   lane = slot-in-step * 16 + block * 4 + column, one 4x4 product per block of four samples; the same four
//   columns are replicated over the four blocks):  x: vech columns 208, 209 (+ two zero columns);
//   y: M_i columns 16..19.
// 210 + 20 useful columns ride on 14 x 64 + 2 x 16 = 928 MFMA cycles per step instead of 16 x 64 = 1024.
// Padding slots (slot_pixel < 0) are zero rows.  M_dev is dense row-major [n][20].
// ---------------------------------------------------------------------------------------------
#define DLAGP_MFMA_SMALL_W_COL 208  // first vech column carried by the small w-MFMA
#define DLAGP_MFMA_SMALL_Q_COL 16   // first column of M carried by the small q-MFMA

__device__ __forceinline__ void vech_invert(int col, int &a, int &b) {
    a = (int)((sqrt(8.0 * col + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= col) ++a;
    while (a * (a + 1) / 2 > col) --a;
    b = col - a * (a + 1) / 2;
}

// Ranks above 20 (K = the rank, `colgroups` > 1): the same layout once per column group g, group after group; group g
// carries vech columns 210 g .. 210 g + 209 (13 full tiles + the small w-MFMA's two) and columns 20 g .. 20 g + 19 of M (tile
// 13 + the small q-MFMA's four); columns beyond k (k + 1) / 2 resp. k are zero.
__global__ void build_mfma_operand(const double *__restrict__ M_dev, const int *__restrict__ slot_pixel, int n_slots,
                                   double *__restrict__ Pm, int K, int colgroups) {
    const long per_group = (long)n_slots * DLAGP_MFMA_PAIRS * 16 * 2;  // (slot, pair, lane-in-slot 0..15, half)
    const long total = per_group * colgroups;
    const int W = K * (K + 1) / 2;
    for (long at = blockIdx.x * (long)blockDim.x + threadIdx.x; at < total; at += (long)gridDim.x * blockDim.x) {
        const int g = (int)(at / per_group);
        const long idx = at % per_group;
        const int half = idx & 1;
        const int cc = (idx >> 1) & 15;
        const int pair = (idx >> 5) % DLAGP_MFMA_PAIRS;
        const int slot = idx / (32 * DLAGP_MFMA_PAIRS);
        const int i = slot_pixel[slot];
        double val = 0.0;
        if (i >= 0) {
            const double *row = M_dev + (long)i * K;
            int wcol = -1, qcol = -1;  // column of vech(M_i M_i') resp. of M_i this element carries
            if (pair < DLAGP_MFMA_PAIRS - 1) {
                const int tile = 2 * pair + half;
                if (tile < DLAGP_MFMA_W_TILES) wcol = g * DLAGP_MFMA_GROUP_W_COLS + tile * 16 + cc;
                else qcol = g * DLAGP_MFMA_GROUP_Q_COLS + cc;
            } else {
                const int j = cc & 3;  // replicated over the four blocks cc >> 2
                if (half == 0) {
                    if (j < 2) wcol = g * DLAGP_MFMA_GROUP_W_COLS + DLAGP_MFMA_SMALL_W_COL + j;
                } else {
                    qcol = g * DLAGP_MFMA_GROUP_Q_COLS + DLAGP_MFMA_SMALL_Q_COL + j;
                }
            }
            if (wcol >= 0 && wcol < W) {
                int a, b;
                vech_invert(wcol, a, b);
                val = row[a] * row[b];
            } else if (qcol >= 0 && qcol < K) {
                val = row[qcol];
            }
        }
        const int step = slot / DLAGP_STEP_PIXELS, s_in = slot % DLAGP_STEP_PIXELS;
        Pm[g * per_group + (((long)step * DLAGP_MFMA_PAIRS + pair) * 64 + s_in * 16 + cc) * 2 + half] = val;
    }
}

// per-slot records {y, mu, omega2, v, window start, 3 spare} streamed by the MFMA kernel; padding slots
// (slot_pixel < 0) carry y = mu = omega2 = 0, v = 1: d = 1, log d = 0, r = 0, zero operand rows
__global__ void fill_records(int n_slots, const int *__restrict__ slot_pixel, const int *__restrict__ slot_pos,
                             const double *__restrict__ y, const double *__restrict__ mu,
                             const double *__restrict__ omega2, const double *__restrict__ v,
                             double *__restrict__ aux) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const int i = slot_pixel[s];
    double *rec = aux + (long)s * 8;
    rec[0] = i >= 0 ? y[i] : 0.0;
    rec[1] = i >= 0 ? mu[i] : 0.0;
    rec[2] = i >= 0 ? omega2[i] : 0.0;
    rec[3] = i >= 0 ? v[i] : 1.0;
    rec[4] = (double)slot_pos[s];
    rec[5] = rec[6] = rec[7] = 0.0;
}

// ---------------------------------------------------------------------------------------------
// Finalisation: one thread per sample.
//   partials: part[split][block][column][16] (partial_index), row = sightline * S + sample
//   B = I + sum_splits (w-tiles),  u = sum_splits (q-tiles),
//   L L' = B (lower, null_gp.py:343), t = L^-1 u,
//   log p = -1/2 ( sum r^2/d - t't + sum log d + 2 sum log L_ii + n log 2pi )  (null_gp.py:354-358)
// The reference forms K^-1 y through two explicit triangular inverses (:346-347); t't is the
// same quantity y' D^-1 M B^-1 M' D^-1 y.  Splits are summed in index order (deterministic).
// A non-positive pivot yields NaN and raises *status (the reference raises LinAlgError);
// NaN inputs propagate to a NaN result without raising, as in the reference.
// KT > 0: compile-time rank, everything in registers.  KT == 0: runtime rank, local arrays.
// ---------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(DLAGP_FINALIZE_THREADS) void finalize_samples(const double *__restrict__ part, int nsplit, long rows, long S,
                                 const SightlineDev *__restrict__ sightlines,
                                 double *__restrict__ out, int *__restrict__ status,
                                 const unsigned char *__restrict__ part_counts) {
    const long row = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const long b = row / S, smp = row % S;
    const SightlineDev &sl = sightlines[b];
    const int k = KT > 0 ? KT : sl.k;
    const int rw = sl.row_width;
    const int qbase = sl.ntw * DLAGP_TILE_COLS;
    const int sbase = (sl.ntw + sl.ntq) * DLAGP_TILE_COLS;
    const long ngroups = (S + 15) / 16;
    const long nblocks = (rows / S) * ngroups;
    const long block = b * ngroups + smp / 16;
    const int lane16 = (int)(smp % 16);
    if (part_counts) nsplit = part_counts[block];  // pieces this block of 16 samples was cut into
    auto partial = [&](int h, int col) { return part[partial_index(h, nblocks, block, rw, col, lane16)]; };

    constexpr int KMAX = KT > 0 ? KT : DLAGP_MAX_RANK;
    // fixed rank: unroll everything so L and t live in registers; runtime rank: keep the loops
    constexpr int UNROLL = KT > 0 ? 4096 : 1;
    double L[KMAX * (KMAX + 1) / 2];
    double t[KMAX];

    double sum_r2 = 0.0, sum_logd = 0.0;
#pragma unroll UNROLL
    for (int c = 0; c < KMAX * (KMAX + 1) / 2; ++c) {
        if (c < k * (k + 1) / 2) {
            double acc = 0.0;
            for (int h = 0; h < nsplit; ++h) acc += partial(h, c);
            L[c] = acc;
        }
    }
#pragma unroll UNROLL
    for (int c = 0; c < KMAX; ++c) {
        if (c < k) {
            double acc = 0.0;
            for (int h = 0; h < nsplit; ++h) acc += partial(h, qbase + c);
            t[c] = acc;
        }
    }
    for (int h = 0; h < nsplit; ++h) {
        sum_r2 += partial(h, sbase);
        sum_logd += partial(h, sbase + 1);
    }

    bool bad = false;
    double tt = 0.0, log_diag = 0.0;
#pragma unroll UNROLL
    for (int i = 0; i < KMAX; ++i) {
        if (i < k) {
#pragma unroll UNROLL
            for (int j = 0; j <= i; ++j) {
                double s = L[vech_col(i, j)];
                if (j == i) s += 1.0;  // identity on the diagonal (null_gp.py:341)
#pragma unroll UNROLL
                for (int m = 0; m < j; ++m) s = fma(-L[vech_col(i, m)], L[vech_col(j, m)], s);
                if (j == i) {
                    if (s <= 0.0) bad = true;  // NaN pivots propagate silently, as LAPACK's do
                    const double diag = sqrt(s);
                    L[vech_col(i, i)] = diag;
                    log_diag += log(diag);
                    double ti = t[i];
#pragma unroll UNROLL
                    for (int m = 0; m < i; ++m) ti = fma(-L[vech_col(i, m)], t[m], ti);
                    ti /= diag;
                    t[i] = ti;
                    tt = fma(ti, ti, tt);
                } else {
                    L[vech_col(i, j)] = s / L[vech_col(j, j)];
                }
            }
        }
    }
    double result = -0.5 * ((sum_r2 - tt) + (sum_logd + 2.0 * log_diag) + sl.n * DLAGP_LOG_2PI);  // one sightline per row here
    if (bad) {
        result = __builtin_nan("");
        atomicOr(status, 1);
    }
    out[sl.batch_index * S + smp] = result;  // the caller's row of this sightline
}

// ---------------------------------------------------------------------------------------------
// Finalisation for ranks above 20: ONE WAVE per sample (finalize_samples<0> gives a sample to one thread, whose
// k (k + 1) / 2 matrix entries then live in scratch memory: 13.7 ms for 10 000 samples at k = 64, four times the
// accumulate kernel).  The sample's matrix B = I + [sum of the pieces] sits packed in LDS (row i at i (i + 1) / 2) with u
// behind it; column by column (right-looking): lane = row scales its entry of column j, takes its share of the forward
// solve, and updates its row's trailing entries.  Same formula as null_gp.py:343-358, another summation order than
// finalize_samples (agreement to rounding).  blockDim = 64 x DLAGP_FINW_SAMPLES; dynamic LDS: that many matrices.
// ---------------------------------------------------------------------------------------------
// one wave's LDS writes before another lane's reads: the LDS serves a wave's operations in issue order, so program order
// (the asm's memory clobber) and an lgkmcnt wait are all it takes
__device__ __forceinline__ void finw_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64 * DLAGP_FINW_SAMPLES) void finalize_wave(
    const double *__restrict__ part, int nsplit, long rows, long S, const SightlineDev *__restrict__ sightlines,
    double *__restrict__ out, int *__restrict__ status, const unsigned char *__restrict__ part_counts) {
    extern __shared__ double finw_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = blockIdx.x * (long)DLAGP_FINW_SAMPLES + wave;
    if (row >= rows) return;  // whole waves leave; no workgroup barrier below
    const long b = row / S, smp = row % S;
    const SightlineDev &sl = sightlines[b];
    const int k = sl.k, W = k * (k + 1) / 2;
    const int rw = sl.row_width;
    const int qbase = sl.ntw * DLAGP_TILE_COLS, sbase = (sl.ntw + sl.ntq) * DLAGP_TILE_COLS;
    const long ngroups = (S + 15) / 16;
    const long nblocks = (rows / S) * ngroups;
    const long block = b * ngroups + smp / 16;
    const int lane16 = (int)(smp % 16);
    if (part_counts) nsplit = part_counts[block];
    double *A = finw_lds + (long)wave * finalize_wave_lds_doubles(DLAGP_MAX_RANK);
    double *u = A + W;
    auto partial = [&](int h, int col) { return part[partial_index(h, nblocks, block, rw, col, lane16)]; };
    for (int c = lane; c < W + k; c += 64) {
        double acc = 0.0;
        for (int h = 0; h < nsplit; ++h) acc += partial(h, c < W ? c : qbase + (c - W));
        A[c] = acc;
    }
    double sum_r2 = 0.0, sum_logd = 0.0;
    for (int h = 0; h < nsplit; ++h) {
        sum_r2 += partial(h, sbase);
        sum_logd += partial(h, sbase + 1);
    }
    finw_sync();
    bool bad = false;
    double tt = 0.0, log_diag = 0.0;
    for (int j = 0; j < k; ++j) {
        const double s = A[vech_col(j, j)] + 1.0;  // identity on the diagonal (null_gp.py:341); already updated by columns < j
        if (s <= 0.0) bad = true;                  // NaN pivots propagate silently, as LAPACK's do
        const double diag = sqrt(s);
        log_diag += log(diag);
        const double tj = u[j] / diag;  // forward solve, column-oriented: u_i -= L_ij t_j for i > j
        tt = fma(tj, tj, tt);
        const int i = j + 1 + lane;  // this lane's row (k <= 64: one row per lane)
        double lij = 0.0;
        if (i < k) {
            lij = A[vech_col(i, j)] / diag;
            A[vech_col(i, j)] = lij;
            u[i] = fma(-lij, tj, u[i]);
        }
        finw_sync();
        if (i < k) {
            // trailing update of row i: A[i][m] -= L[i][j] L[m][j], m = j + 1 .. i
            double *row_i = A + vech_col(i, 0);
            for (int m = j + 1; m <= i; ++m) row_i[m] = fma(-lij, A[vech_col(m, j)], row_i[m]);
        }
        finw_sync();
    }
    if (lane == 0) {
        double result = -0.5 * ((sum_r2 - tt) + (sum_logd + 2.0 * log_diag) + sl.n * DLAGP_LOG_2PI);
        if (bad) {
            result = __builtin_nan("");
            atomicOr(status, 1);
        }
        out[sl.batch_index * S + smp] = result;  // the caller's row of this sightline
    }
}

// ---------------------------------------------------------------------------------------------
// Finalisation for the fixed rank K = 20 (the production path): 16 samples per workgroup.
//   stage   every column of [B - I | u | scalars], summed over the splits in index order, goes to
//           LDS as lds[column][sample]: independent coalesced loads shared by the 4 waves of the
//           workgroup, many in flight (a register-resident one-thread-per-sample variant
//           serialised on them)
//   factor  row-by-row Cholesky (Cholesky-Banachiewicz): row i is finished in registers from the
//           rows before it, which are read back from LDS; pivots are inverted once (20 rcp instead
//           of 190 divisions) and log det B = log prod_i L_ii^2 costs a single log
//   solve   t_i = (u_i - sum_m L_im t_m) / L_ii interleaved with the factorisation
// Same results as finalize_samples<20> to rounding; ~15x faster on 10 000 samples.
// Dynamic LDS: DLAGP_FINALIZE_LDS_DOUBLES doubles.
// ---------------------------------------------------------------------------------------------
// launches of up to this many blocks of 16 samples factor with 16 lanes per sample, larger ones with one lane per
// sample (option finalize_row16_blocks).  The 16-lane form was written for small launches (few workgroups per CU: the
// dependent chain of the one-lane form is what they wait for) but is the faster one on large ones too -- 442 against
// 471 us on 40 000 blocks, 240 against 251 us on 20 000 (profiles/r03_experiments.txt item 18) -- so the default is "all".

// f(integral_constant<int, I>) for I = BEGIN .. END - 1, unrolled at compile time (DPP controls and register indices
// must be constants)
template <int BEGIN, int END, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (BEGIN < END) {
        f(std::integral_constant<int, BEGIN>{});
        static_for<BEGIN + 1, END>(f);
    }
}

// Where the MFMA kernel leaves element (sample, column) of a block's partial sums (a block-piece is row_width x 16
// doubles, as in the column-major layout of the generic kernel): its 14 full tiles as the MFMA hands them out, two
// accumulator registers per store -- [tile][r >> 1][lane][r & 1], lane = (sample & 3) * 16 + column-in-tile, r = sample >> 2 --
// so that the kernel's epilogue is 28 stores of 1 KB (16 bytes per lane: stores are issue-bound, MI355X_MICROARCH.md) and
// no transposition through LDS; then the nine extras, 16 samples each: vech columns 208, 209, u 16..19 (the 4x4x4
// MFMAs' outputs), sum r^2/d, sum log d, n log 2pi.
#define DLAGP_NATIVE_EXTRAS (DLAGP_MFMA_BIG_TILES * 256)
__host__ __device__ __forceinline__ int native_tile_offset(int tile, int sample, int cc) {
    const int r = sample >> 2;
    return ((tile * 2 + (r >> 1)) * 64 + ((sample & 3) << 4) + cc) * 2 + (r & 1);
}

// Factorisation + forward solve of ONE sample's B = I + sum_p w_p M_p M_p' (null_gp.py:339-356) by the SIXTEEN lanes of a
// DPP row: the body of finalize_rank20<ROW16 = true, ..>, shared with the accumulate kernel's own finalisation of
// single-piece blocks (kernels_mfma.h, FUSED).  `col`: the sample's staged columns in LDS, column c at col[c * LS] --
// c < 210: vech(B - I); 210..229: u; 230: sum r^2/d; 231: sum log d.  Lane q of the row holds row q of the matrix (and,
// q <= 4, row 16 + q; u rides along as row 20, so its row of the factor is t = L^-1 u).  Returns log p in lane q == 4 of
// the row (other lanes: scratch); `bad`: a pivot was not positive (B not positive definite).  Every entry sees the same
// operations in the same order wherever this is called from: the two finalisations agree bit for bit.
template <int LS>
__device__ __forceinline__ double factor_solve_row16(const double *col, int q, double n_log_2pi, bool &bad) {
    constexpr int K = DLAGP_FIN_K;
    constexpr int NC = DLAGP_FIN_COLS;
        double A0[16], A1[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            // row i < 20: vech(B - I) entry (i, k), the identity added on the diagonal (null_gp.py:341); row 20: u; columns
            // beyond the row and rows 21..31 start at zero.  Branch-free: a clamped, always valid address and a select.
            if (k < 16) {
                const int kc = k < q ? k : q;
                double v = col[(q * (q + 1) / 2 + kc) * LS];
                v = k <= q ? v : 0.0;
                if (k == q) v += 1.0;
                A0[k] = v;
            }
            const int i = 16 + q, ic = i < K ? i : K;
            const int kc = ic < K ? (k < ic ? k : ic) : k;
            double v = col[(ic < K ? ic * (ic + 1) / 2 + kc : NC + k) * LS];
            v = (i < K ? k <= i : i == K) ? v : 0.0;
            if (k == i) v += 1.0;
            A1[k] = v;
        }
        LogProduct pivots;
        // DPP row_newbcast: lane L of every row of 16 lanes, as ONE 64-bit move (v_mov_b64_dpp, gfx90a and later; the 32-bit
        // halves cost two).  The DPP instructions of this routine are inline assembly in program order (volatile): the
        // compiler's hazard recogniser does not see inside them, and a DPP read of a VGPR needs two wait states behind the
        // VALU write of it -- so every place where the source may just have been written carries its own s_nop 1: this
        // broadcast (its source was written by the previous column's updates) and the first update of a column (its
        // sources were scaled right before it).
        auto row_bcast = [](double v, auto lane_c) __attribute__((always_inline)) {
            double out;
            asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                         : "=v"(out) : "v"(v), "n"(decltype(lane_c)::value));
            return out;
        };
        // fully unrolled: every register index and DPP control below is a compile-time constant
        auto column = [&](auto j_c) __attribute__((always_inline)) {
            constexpr int j = decltype(j_c)::value;
            double s;
            if constexpr (j < 16) s = row_bcast(A0[j], std::integral_constant<int, j>{});
            else s = row_bcast(A1[j], std::integral_constant<int, j - 16>{});
            if (s <= 0.0) bad = true;  // NaN pivots propagate silently, as LAPACK's do
            pivots.multiply(s);        // log det B = sum log L_ii^2
            const double inv = fast_rsqrt(s);
            if constexpr (j < 16) A0[j] *= inv;  // rows > j: L_ij (row j itself and the rows above it: scratch)
            A1[j] *= inv;
            // Trailing update of column k: entry (row, k) -= L(row, j) L(k, j), where L(k, j) sits in lane k (mod 16) of the
            // row of 16 lanes.  The broadcast rides in the multiply-add itself: on gfx90a and later the FP64 VOP2 forms take a
            // DPP row_newbcast source (v_fmac_f64_dpp: D += bcast(src0) * src1), so a (j, k) pair costs two instructions
            // instead of two 32-bit DPP moves + two FMAs -- a third of this routine's instructions, which a single wave
            // issues every 8 cycles (profiles/r05_experiments.txt item 7).  Same products, same rounding: c - l a either way.
            auto update = [&](auto k_c) __attribute__((always_inline)) {
                constexpr int k = decltype(k_c)::value;
                constexpr int lane = k < 16 ? k : k - 16;
                const double src = k < 16 ? A0[j] : A1[j];  // L(k, j) lives in lane `lane` of this register
                if constexpr (k == j + 1) {  // the column's first DPP read of the freshly scaled sources: wait states first
                    if constexpr (k < 16)
                        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                                     : "+v"(A0[k]) : "v"(src), "v"(A0[j]), "n"(lane));
                    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                                 : "+v"(A1[k]) : "v"(src), "v"(A1[j]), "n"(lane));
                } else {
                    if constexpr (k < 16)
                        asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                                     : "+v"(A0[k]) : "v"(src), "v"(A0[j]), "n"(lane));
                    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                                 : "+v"(A1[k]) : "v"(src), "v"(A1[j]), "n"(lane));
                }
            };
            static_for<j + 1, K>(update);
            __builtin_amdgcn_sched_barrier(0);  // columns in program order (the scheduler otherwise interleaves several)
        };
        static_for<0, K>(column);
        double tt = 0.0;
#pragma unroll
        for (int m = 0; m < K; ++m) tt = fma(A1[m], A1[m], tt);
        const double sum_r2 = col[(NC + K) * LS];
        const double sum_logd = col[(NC + K + 1) * LS];
        double result = -0.5 * ((sum_r2 - tt) + (sum_logd + pivots.log_value()) + n_log_2pi);
        return result;
}

// Ranks above 20 on the MFMA path.  Column group g of a (piece, block) leaves its sums in the native layout above, at
// piece index g nparts + piece of a buffer of DLAGP_MFMA_ROW_WIDTH-wide rows.  This kernel adds a block's pieces in piece
// order and sorts the groups' columns into ONE column-major row of the sightline's own width -- the generic kernel's
// layout -- for finalize_samples: vech column c of group c / 210, column j of u of group j / 20, the three scalars of
// group 0 (every group forms them; they are equal).  grid: blocks of 16 samples.
__global__ __launch_bounds__(256) void regroup_native(const double *__restrict__ part, int nparts, long S,
                                                      const SightlineDev *__restrict__ sightlines,
                                                      const unsigned char *__restrict__ part_counts,
                                                      double *__restrict__ rows_out) {
    const long block = blockIdx.x, nblocks = gridDim.x;
    const long ngroups = (S + 15) / 16;
    const SightlineDev &sl = sightlines[block / ngroups];
    const int k = sl.k, W = k * (k + 1) / 2;
    const int qbase = sl.ntw * DLAGP_TILE_COLS, sbase = (sl.ntw + sl.ntq) * DLAGP_TILE_COLS;
    const int nsplit = part_counts ? part_counts[block] : nparts;
    constexpr int RW = (14 + 2) * DLAGP_TILE_COLS + DLAGP_ROW_EXTRA;  // DLAGP_MFMA_ROW_WIDTH (kernels_mfma.h): a rank-20 row
    for (int idx = threadIdx.x; idx < (W + k + 3) * 16; idx += blockDim.x) {
        const int c = idx >> 4, s = idx & 15;
        int g, off, dest;
        if (c < W) {
            g = c / DLAGP_MFMA_GROUP_W_COLS;
            const int l = c % DLAGP_MFMA_GROUP_W_COLS;
            off = l < DLAGP_MFMA_SMALL_W_COL ? native_tile_offset(l >> 4, s, l & 15)
                                             : DLAGP_NATIVE_EXTRAS + (l - DLAGP_MFMA_SMALL_W_COL) * 16 + s;
            dest = c;
        } else if (c < W + k) {
            const int j = c - W;
            g = j / DLAGP_MFMA_GROUP_Q_COLS;
            const int l = j % DLAGP_MFMA_GROUP_Q_COLS;
            off = l < DLAGP_MFMA_SMALL_Q_COL ? native_tile_offset(DLAGP_MFMA_W_TILES, s, l)
                                             : DLAGP_NATIVE_EXTRAS + (2 + l - DLAGP_MFMA_SMALL_Q_COL) * 16 + s;
            dest = qbase + j;
        } else {
            g = 0;
            off = DLAGP_NATIVE_EXTRAS + (6 + c - W - k) * 16 + s;
            dest = sbase + (c - W - k);
        }
        double acc = 0.0;
        for (int h = 0; h < nsplit; ++h) acc += part[partial_index((long)g * nparts + h, nblocks, block, RW, 0, 0) + off];
        rows_out[partial_index(0, nblocks, block, sl.row_width, dest, s)] = acc;
    }
}

// The tail of finalize_rank20 with an EvidenceTail (dlagp_device.h): called by ALL threads of the workgroup, uniformly.
//   writer / smp / value: this thread holds sample `smp`'s log-density (NaN if its matrix was not positive definite)
//   row: the sample's position in the caller's arrays; b / g / ngroups: sightline, block of the sightline, blocks per sightline
// The blocks' triples are combined in an order that does not depend on which workgroup arrives last: thread t of the last
// workgroup takes blocks t, t + 256, ... in that order, the threads' sums meet in a fixed tree -- the same bits run after run.
__device__ __forceinline__ double tail_load(const double *p) {  // from the device's coherent level, never a stale cache line
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void evidence_tail(const EvidenceTail &ev, bool writer, int smp, double value, long row, double *out,
                                              long b, long g, long ngroups, int batch_index, int valid) {
    __shared__ double tail_v[DLAGP_FIN_SAMPLES];
    __shared__ double tail_red[3 * (DLAGP_FIN_THREADS / 64)];
    __shared__ int tail_last;
    const int tid = threadIdx.x;
    const double nan = __builtin_nan("");
    if (writer) {
        double v = value - ev.log_s;  // the "additional Occam's razor" (dla_gp.py:157-159)
        if (ev.too_close && ev.too_close[row]) v = nan;  // separation mask (:164-177)
        out[row] = v;
        tail_v[smp] = v;
    }
    __syncthreads();
    if (tid < 64) {  // the block's triple: max, sum of exp(. - max) in sample order, count of finite samples
        const double x = tid < valid ? tail_v[tid] : nan;  // (the block's samples beyond S have no writer)
        double m = x == x ? x : -__builtin_inf();
        for (int off = 8; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
        const double e = x == x ? exp(x - m) : 0.0;
        double sum = 0.0, count = 0.0;
        for (int i = 0; i < DLAGP_FIN_SAMPLES; ++i) {
            sum += __shfl(e, i);
            count += __shfl(x == x ? 1.0 : 0.0, i);
        }
        if (tid == 0) {
            double *mine = ev.part + 3 * (b * ngroups + g);
            __hip_atomic_store(mine + 0, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 1, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 2, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // The hand-off of MI355X_MICROARCH.md ("hand-offs measured with sc1 loads in place of the acquire", first row):
            // write-through (sc1) stores, this wave's wait for them, then ONE relaxed agent-scope add; the workgroup whose add
            // came last reads with sc1 loads behind a barrier.  No agent-scope fence: a release per workgroup writes back the
            // XCD's L2 (1.7-6.5 us each, four workgroups per compute unit) -- with __threadfence() here this kernel took
            // 40 us instead of 18 and the fold LOST 22 us per launch (profiles/r06_experiments.txt item 16).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tail_last = __hip_atomic_fetch_add(ev.tickets + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)ngroups - 1;
        }
    }
    __syncthreads();
    if (!tail_last) return;  // (the barrier above is the one the adding wave joins before anybody loads)
    const double *parts = ev.part + 3 * b * ngroups;
    const int wave = tid >> 6;
    // this thread's blocks t, t + 256, ...: all of their triples requested before the first is used (up to KEEP blocks per
    // thread stay in registers: 16 384 samples per sightline; longer levels read theirs again)
    constexpr int KEEP = 4;
    double pm[KEEP], ps[KEEP], pc[KEEP];
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const long blk = tid + (long)k * DLAGP_FIN_THREADS;
        const bool there = blk < ngroups;
        pm[k] = there ? tail_load(parts + 3 * blk) : 0.0;
        ps[k] = there ? tail_load(parts + 3 * blk + 1) : 0.0;
        pc[k] = there ? tail_load(parts + 3 * blk + 2) : 0.0;
    }
    double m = -__builtin_inf();
#pragma unroll
    for (int k = 0; k < KEEP; ++k)
        if (pc[k] > 0.0) m = fmax(m, pm[k]);
    for (long blk = tid + (long)KEEP * DLAGP_FIN_THREADS; blk < ngroups; blk += DLAGP_FIN_THREADS)
        if (tail_load(parts + 3 * blk + 2) > 0.0) m = fmax(m, tail_load(parts + 3 * blk));
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((tid & 63) == 0) tail_red[wave] = m;
    __syncthreads();
    m = tail_red[0];
    for (int w = 1; w < DLAGP_FIN_THREADS / 64; ++w) m = fmax(m, tail_red[w]);
    __syncthreads();
    double sum = 0.0, count = 0.0;
#pragma unroll
    for (int k = 0; k < KEEP; ++k)
        if (pc[k] > 0.0) {
            sum += ps[k] * exp(pm[k] - m);
            count += pc[k];
        }
    for (long blk = tid + (long)KEEP * DLAGP_FIN_THREADS; blk < ngroups; blk += DLAGP_FIN_THREADS) {
        const double c = tail_load(parts + 3 * blk + 2);
        if (c > 0.0) {
            sum += tail_load(parts + 3 * blk + 1) * exp(tail_load(parts + 3 * blk) - m);
            count += c;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off);
        count += __shfl_xor(count, off);
    }
    if ((tid & 63) == 0) {
        tail_red[wave] = sum;
        tail_red[DLAGP_FIN_THREADS / 64 + wave] = count;
    }
    __syncthreads();
    if (tid == 0) {
        double total = 0.0, n = 0.0;
        for (int w = 0; w < DLAGP_FIN_THREADS / 64; ++w) {
            total += tail_red[w];
            n += tail_red[DLAGP_FIN_THREADS / 64 + w];
        }
        // an all-NaN level: nanmax and nanmean are NaN (dla_gp.py:180-190)
        ev.evidence[batch_index] = n > 0.0 ? m + log(total / n) - ev.level * ev.log_s : nan;
        ev.tickets[b] = 0;  // the next launch finds its ticket counter at zero
    }
}

// NATIVE: the partial sums come from the MFMA kernel (layout above); else from the generic kernel (column-major).
template <bool ROW16, bool NATIVE>
__global__ __launch_bounds__(DLAGP_FIN_THREADS) void finalize_rank20(const double *__restrict__ part, int nsplit, long rows, long S,
                                                      const SightlineDev *__restrict__ sightlines,
                                                      double *__restrict__ out, int *__restrict__ status,
                                                      const unsigned char *__restrict__ part_counts,
                                                      double *__restrict__ out2 = nullptr, long pair_S = 0,
                                                      EvidenceTail ev = EvidenceTail{}) {
    extern __shared__ double fin_lds[];
    constexpr int K = DLAGP_FIN_K;
    constexpr int NC = DLAGP_FIN_COLS;
    constexpr int NS = DLAGP_FIN_SAMPLES;
    const int tid = threadIdx.x;
    const long ngroups = (S + NS - 1) / NS;
    const long nblocks = (rows / S) * ngroups;
    const long block = blockIdx.x;
    if (part_counts) nsplit = part_counts[block];  // pieces this block's pixels were cut into (work items, dlagp_hip.hip)
    const long b = block / ngroups, g = block % ngroups;
    const long row0 = sightlines[b].batch_index * S + g * NS;  // first output row of this workgroup (the caller's arrays)
    const int valid = (int)((S - g * NS) < NS ? (S - g * NS) : NS);
    // Pair launch (SampleSource::pair_S, dlagp_device.h): the block's rows 0..7 are points 8 g .. 8 g + 7 of the first model
    // (-> out), rows 8..15 the same points of the second (-> out2); S is then the launch's virtual sample count.
    auto store = [&](int smp, double result) __attribute__((always_inline)) {
        if (pair_S) {
            const long point = g * 8 + (smp & 7);
            if (point < pair_S) (smp >= 8 ? out2 : out)[sightlines[b].batch_index * pair_S + point] = result;
        } else if (smp < valid) {
            out[row0 + smp] = result;
        }
    };
    const SightlineDev &sl = sightlines[b];
    const int rw = sl.row_width;
    const int qbase = sl.ntw * DLAGP_TILE_COLS;
    const int sbase = (sl.ntw + sl.ntq) * DLAGP_TILE_COLS;

    // ---- stage: lds[c][sample], c < NC: vech(B - I); NC..NC+K-1: u; NC+K: sum r^2/d; NC+K+1: sum log d;
    // NC+K+2: n log 2pi.  One contiguous 33 KB block per split: fully coalesced reads, which are what
    // this kernel costs (a CU sustains only tens of GB/s from HBM/MALL, hence also the small workgroups).
    // All loads of a thread are issued before the first use (the loop is fully unrolled for the usual
    // split counts): issued one iteration at a time, the ~15 dependent round trips to L2/HBM were 70 % of
    // this kernel.
    // 16-byte loads: a thread takes two neighbouring samples of a column (the 16 samples of a column are 128
    // contiguous bytes), which halves the number of load instructions in flight per byte.
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int LS = DLAGP_FIN_LDS_STRIDE;
    // One staging item = 16 bytes of a block-piece and the two LDS slots they go to.
    //   column-major layout: two neighbouring samples of one staged column (236 x 8 items);
    //   native layout: one lane's register pair of a tile -- samples s and s + 4 of one column (14 x 2 x 64 items, in
    //   memory order: perfectly coalesced), then the extras two samples at a time (9 x 8 items).
    constexpr int TILE_ITEMS = DLAGP_MFMA_BIG_TILES * 128;
    constexpr int PAIRS = NATIVE ? TILE_ITEMS + 9 * (NS / 2) : DLAGP_FIN_STAGED * (NS / 2);
    constexpr int ITER = (PAIRS + DLAGP_FIN_THREADS - 1) / DLAGP_FIN_THREADS;
    const long piece_stride = nblocks * (long)rw * 16;
    const double *piece0 = part + partial_index(0, nblocks, block, rw, 0, 0);
    auto locate = [&](int idx, int &src, int &lds_a, int &lds_b) __attribute__((always_inline)) {
        if constexpr (NATIVE) {
            if (idx < TILE_ITEMS) {
                const int cc = idx & 15, q = (idx >> 4) & 3, rp = (idx >> 6) & 1, t = idx >> 7;
                const int c = (t < DLAGP_MFMA_W_TILES ? 16 * t : NC) + cc;  // tile 13 holds u 0..15
                src = 2 * idx;
                lds_a = c * LS + q + 8 * rp;  // sample q + 4 (2 rp)
                lds_b = lds_a + 4;            // sample q + 4 (2 rp + 1)
            } else {
                const int e = (idx - TILE_ITEMS) >> 3, sp = 2 * ((idx - TILE_ITEMS) & 7);
                const int c = e < 2 ? DLAGP_MFMA_SMALL_W_COL + e : (e < 6 ? NC + DLAGP_MFMA_SMALL_Q_COL + (e - 2) : NC + K + (e - 6));
                src = DLAGP_NATIVE_EXTRAS + e * 16 + sp;
                lds_a = c * LS + sp;
                lds_b = lds_a + 1;
            }
        } else {
            const int c = idx / (NS / 2), smp = 2 * (idx % (NS / 2));
            src = (c < NC ? c : (c < NC + K ? qbase + (c - NC) : sbase + (c - NC - K))) * 16 + smp;
            lds_a = c * LS + smp;
            lds_b = lds_a + 1;
        }
    };
    auto stage = [&](auto nsplit_c) {
        constexpr int H = decltype(nsplit_c)::value;
        d2 v[ITER][H];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int idx = tid + it * DLAGP_FIN_THREADS;
            int src, lds_a, lds_b;
            locate(idx < PAIRS ? idx : tid, src, lds_a, lds_b);  // clamp: a valid address, value unused
#pragma unroll
            for (int h = 0; h < H; ++h) v[it][h] = *reinterpret_cast<const d2 *>(piece0 + h * piece_stride + src);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int idx = tid + it * DLAGP_FIN_THREADS;
            d2 acc = d2{0.0, 0.0};
#pragma unroll
            for (int h = 0; h < H; ++h) acc += v[it][h];
            int src, lds_a, lds_b;
            locate(idx < PAIRS ? idx : tid, src, lds_a, lds_b);
            if (idx < PAIRS) {
                fin_lds[lds_a] = acc.x;
                fin_lds[lds_b] = acc.y;
            }
        }
    };
    if (nsplit == 1) stage(std::integral_constant<int, 1>{});
    else if (nsplit == 2) stage(std::integral_constant<int, 2>{});
    else if (nsplit == 3) stage(std::integral_constant<int, 3>{});
    else if (nsplit == 4) stage(std::integral_constant<int, 4>{});
    else if (nsplit == 5) stage(std::integral_constant<int, 5>{});
    else {
        // more pieces (a launch of few blocks cut into up to 32 pieces each, dlagp_plan_cut): four pieces' loads in flight
        // at a time, added in piece order like the unrolled forms above
        d2 acc[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) acc[it] = d2{0.0, 0.0};
        for (int h0 = 0; h0 < nsplit; h0 += 4) {
            d2 v[ITER][4];
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int idx = tid + it * DLAGP_FIN_THREADS;
                int src, lds_a, lds_b;
                locate(idx < PAIRS ? idx : tid, src, lds_a, lds_b);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int h = h0 + u < nsplit ? h0 + u : nsplit - 1;  // clamp: a valid address, value unused
                    v[it][u] = *reinterpret_cast<const d2 *>(piece0 + h * piece_stride + src);
                }
            }
#pragma unroll
            for (int it = 0; it < ITER; ++it)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (h0 + u < nsplit) acc[it] += v[it][u];
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int idx = tid + it * DLAGP_FIN_THREADS;
            int src, lds_a, lds_b;
            locate(idx < PAIRS ? idx : tid, src, lds_a, lds_b);
            if (idx < PAIRS) {
                fin_lds[lds_a] = acc[it].x;
                fin_lds[lds_b] = acc[it].y;
            }
        }
    }
    __syncthreads();
    if constexpr (ROW16) {
    // ---- factor + solve: SIXTEEN lanes per sample, all four waves (4 samples each).
    // Lane q of a sample's row of 16 lanes holds row q of the sample's matrix (slot 0) and, for q <= 4, row 16 + q
    // (slot 1) in registers; u rides along as row 20, so that its row of the factor is t = L^-1 u.  Right-looking: for
    // column j the pivot and then the column's entries are handed round the 16 lanes with DPP row broadcasts (no LDS,
    // no waits), every lane scales its own rows' entries and updates their trailing columns.  Every real entry sees
    // the same operations in the same order as in the row-by-row form below (one lane per sample: a dependent chain
    // of 1540 FMAs that took 17 of this kernel's 29 us on a 625-block launch) -- identical results.  Entries above
    // the diagonal and rows 21..31 are scratch: updated like everything else, never read by a real entry.
        const int smp = tid >> 4, q = tid & 15;
        const double *col = fin_lds + smp;
        bool bad = false;
        double result = factor_solve_row16<LS>(col, q, col[(NC + K + 2) * LS], bad);
        const bool writer = q == 4 && smp < valid;  // row 20 = t sits in lane 4; samples beyond S hold unwritten partials
        if (writer && bad) {
            result = __builtin_nan("");
            atomicOr(status, 1);
        }
        if (ev.evidence) {  // (never with a pair launch: the host passes no tail then)
            evidence_tail(ev, writer, smp, result, row0 + smp, out, b, g, ngroups, sightlines[b].batch_index, valid);
            return;
        }
        if (writer) store(smp, result);
        return;
    }
    // ---- factor + solve, one lane per sample (launches of many blocks: fewest instructions per sample)
    const bool active = tid < valid;  // lanes beyond the sightline's last sample hold unwritten partials
    if (!active && !ev.evidence) return;  // (with a tail they take part in its barriers)
    double result = 0.0;
    if (active) {
        double *mine = fin_lds + tid;

        double t[K], inv_diag[K];
        double tt = 0.0;
        LogProduct pivots;
        bool bad = false;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            double r[K];
#pragma unroll
            for (int j = 0; j <= i; ++j) r[j] = mine[vech_col(i, j) * LS];
            r[i] += 1.0;  // identity on the diagonal (null_gp.py:341)
#pragma unroll
            for (int j = 0; j < i; ++j) {
                double s = r[j];
#pragma unroll
                for (int m = 0; m < j; ++m) s = fma(-r[m], mine[vech_col(j, m) * LS], s);
                r[j] = s * inv_diag[j];
            }
            double s = r[i];
#pragma unroll
            for (int m = 0; m < i; ++m) s = fma(-r[m], r[m], s);
            if (s <= 0.0) bad = true;  // NaN pivots propagate silently, as LAPACK's do
            pivots.multiply(s);        // log det B = sum log L_ii^2
            const double inv = fast_rsqrt(s);  // the same pivot arithmetic as the 16-lane form
            inv_diag[i] = inv;
#pragma unroll
            for (int j = 0; j < i; ++j) mine[vech_col(i, j) * LS] = r[j];
            double ti = mine[(NC + i) * LS];
#pragma unroll
            for (int m = 0; m < i; ++m) ti = fma(-r[m], t[m], ti);
            ti *= inv;
            t[i] = ti;
            tt = fma(ti, ti, tt);
        }
        const double sum_r2 = mine[(NC + K) * LS];
        const double sum_logd = mine[(NC + K + 1) * LS];
        const double n_log_2pi = mine[(NC + K + 2) * LS];
        result = -0.5 * ((sum_r2 - tt) + (sum_logd + pivots.log_value()) + n_log_2pi);
        if (bad) {
            result = __builtin_nan("");
            atomicOr(status, 1);
        }
    }
    if (ev.evidence) {
        evidence_tail(ev, active, tid, result, row0 + tid, out, b, g, ngroups, sightlines[b].batch_index, valid);
        return;
    }
    store(tid, result);
}

// ---------------------------------------------------------------------------------------------
// Level evidence (reference dla_gp.py:155-190): one workgroup per sightline.
//   ll[i] <- ll[i] - log S                      (the "additional Occam's razor", :157-159)
//   ll[i] <- NaN where too_close[i]             (separation mask, :164-177; optional)
//   evidence = nanmax + log(nanmean(exp(ll - nanmax))) - level * log S      (:180-190)
// All-NaN levels give NaN, as np.nanmax / np.nanmean do.
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(DLAGP_EVIDENCE_THREADS) void level_evidence(double *__restrict__ ll, long S, int level,
                                                                       const unsigned char *__restrict__ too_close,
                                                                       double *__restrict__ evidence, double log_s) {
    __shared__ double red[DLAGP_EVIDENCE_THREADS / 64];
    __shared__ double red2[DLAGP_EVIDENCE_THREADS / 64];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    double *row = ll + blockIdx.x * S;
    const unsigned char *mask = too_close ? too_close + blockIdx.x * S : nullptr;
    // log_s = log(S) comes from the host: 1024 threads evaluating the same ~400-cycle log was 0.7 us of an 8 us kernel

    // Up to 16 values per thread stay in registers between the two passes, and all of a thread's loads are
    // issued before the first use (a load / store / load chain on the same array serialised on memory
    // latency: 10 us for 10 000 values); longer rows take the looping path.
    constexpr int KEEP = 16;
    const bool in_registers = S <= (long)KEEP * DLAGP_EVIDENCE_THREADS;
    double kept[KEEP];
    double top = -__builtin_inf();
    bool any = false;
    if (in_registers) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const long i = tid + (long)k * DLAGP_EVIDENCE_THREADS;
            kept[k] = i < S ? row[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const long i = tid + (long)k * DLAGP_EVIDENCE_THREADS;
            double v = kept[k] - log_s;
            if (i < S && mask && mask[i]) v = __builtin_nan("");
            if (i >= S) v = __builtin_nan("");
            kept[k] = v;
            if (i < S) row[i] = v;
            if (v == v) {
                top = fmax(top, v);
                any = true;
            }
        }
    } else {
        for (long i = tid; i < S; i += DLAGP_EVIDENCE_THREADS) {
            double v = row[i] - log_s;
            if (mask && mask[i]) v = __builtin_nan("");
            row[i] = v;
            if (v == v) {
                top = fmax(top, v);
                any = true;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) top = fmax(top, __shfl_xor(top, off));
    if ((tid & 63) == 0) red[wave] = top;
    __syncthreads();
    top = red[0];
    for (int w = 1; w < DLAGP_EVIDENCE_THREADS / 64; ++w) top = fmax(top, red[w]);
    __syncthreads();

    double sum = 0.0, count = 0.0;
    if (in_registers) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const double v = kept[k];
            if (v == v) {
                sum += exp(v - top);
                count += 1.0;
            }
        }
    } else {
        for (long i = tid; i < S; i += DLAGP_EVIDENCE_THREADS) {
            const double v = row[i];
            if (v == v) {
                sum += exp(v - top);
                count += 1.0;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off);
        count += __shfl_xor(count, off);
    }
    if ((tid & 63) == 0) {
        red[wave] = sum;
        red2[wave] = count;
    }
    __syncthreads();
    if (tid == 0) {
        double total = 0.0, n = 0.0;
        for (int w = 0; w < DLAGP_EVIDENCE_THREADS / 64; ++w) {
            total += red[w];
            n += red2[w];
        }
        // an all-NaN level: nanmax and nanmean are NaN
        evidence[blockIdx.x] = n > 0.0 ? top + log(total / n) - level * log_s : __builtin_nan("");
    }
    (void)any;
}

// ---------------------------------------------------------------------------------------------
// Level finish: everything the reference does between two occupation levels (dla_gp.py:155-218), no host round trip.
//   1. ll[i] - log S -> sample_ll[b][level][i]; NaN where two absorbers of the sample are closer than
//      min_z_separation (sorted-difference test on the sample's redshifts, :164-177)
//   2. evidence = nanmax + log(nanmean(exp(ll - nanmax))) - level log S (:180-190); a NaN evidence ends
//      the sightline's sweep (:200-206): later levels keep their NaN / zero fill
//   3. unless this is the last level: resample base indices in proportion to W = exp(ll - nanmax) (NaN -> 0)
//      as np.random.choice(S, S, p = W / W.sum()) does -- index = searchsorted(cdf, u, side="right") with the
//      uniforms u drawn on the host from the legacy global RNG, so that the indices are the reference's (:213-218).
//
// Arithmetic, the same in both forms below.  1024 virtual lanes; lane t owns the chunk of samples [t c, (t + 1) c),
// c = ceil(S / 1024).  A lane forms its chunk's values, their maximum m_t, the weights w_i = exp(v_i - m_t) RELATIVE TO ITS
// OWN maximum and their sum s_t in order -- nothing of this waits for another lane.  Then, with M = max m_t and
// f_t = exp(m_t - M): the 64 lanes of a wave scan f_t s_t (Hillis-Steele by shuffles), the 16 waves' totals are added in
// wave order, and the cumulative weight of sample i is its wave's offset + the scan of the lanes before its own + the
// running sum of f_t w_i inside its chunk.  Sample i is drawn for a uniform u when it is the first with
// cum[i] > u cum[S - 1]; evidence = M + log(cum[S - 1] / #finite) - level log S.
// NumPy takes exp(v - M) in one step, normalises and accumulates sequentially: the two cdfs differ by rounding, a few
// 1e-16 (bounded by ~S eps), and an index could differ where a uniform falls that close to a cdf step.  So every search that
// ends within DLAGP_TIE_MARGIN (1e-11, four to five orders above any rounding difference) of the step on either side of its
// answer is COUNTED per (sightline, level) in `ties`, and the host settles a level with a non-zero count with NumPy's own
// arithmetic (sightline.py `settle_near_ties`): the base indices a caller sees are np.random.choice's by construction,
// not with high probability.
//
// Two forms of the same arithmetic (same device functions, same virtual lanes -- identical bits):
//   level_finish_fused    one 1024-thread workgroup per sightline, the two phases separated by a workgroup barrier: 30-40 us
//                         on ONE compute unit whatever the batch size -- a fifth of a one-spectrum model selection's kernel time
//   level_weights + level_resample   two launches that spread a sightline over many compute units: the lanes' chunks by
//                         16 one-wave workgroups (all of a lane's reads in flight at once), the searches by `groups`
//                         workgroups that each rebuild the cumulative array in their LDS (1024 exponentials, S multiply-adds)
//                         and take a share of the uniforms.
// ---------------------------------------------------------------------------------------------
namespace lvl {
__device__ __forceinline__ double *x_m(const LevelArgs &a, long b) { return a.exch + b * DLAGP_LEVEL_EXCH; }
__device__ __forceinline__ double *x_s(const LevelArgs &a, long b) { return x_m(a, b) + DLAGP_LEVEL_LANES; }
__device__ __forceinline__ double *x_cnt(const LevelArgs &a, long b) { return x_s(a, b) + DLAGP_LEVEL_LANES; }
__device__ __forceinline__ double *column(const LevelArgs &a, long b) { return a.sample_ll + (b * a.max_dlas + a.level) * a.S; }

__device__ __forceinline__ double wave_max(double x) {
    for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_xor(x, off));
    return x;
}

// Phase A for virtual lane vt (the calling wave = the 64 consecutive lanes vt & ~63 ..).  Values of the lane's chunk --
// Occam term, separation mask -- into the level's column of sample_ll, BATCH samples at a time with their reads issued
// together: a sample's redshifts sit behind the index table (base -> z: two dependent trips).  "Two absorbers closer than
// the cut" is tested pair by pair: the smallest gap of the sorted redshifts (reference dla_gp.py:164-177: diff of the sort)
// IS the smallest pairwise distance, a NaN redshift compares false either way, and no sorting loop keeps the reads from
// being hoisted.  Then the weights relative to the chunk's own maximum, their sum, the count of finite samples.
template <int BATCH>
__device__ __forceinline__ void lane_weights(const LevelArgs &a, long b, int vt) {
    const long S = a.S;
    const int level = a.level, kd = a.level + 1;
    const double *in = a.ll + b * S;
    double *col = column(a, b);
    double *W = a.weights + b * S;
    const long chunk = (S + DLAGP_LEVEL_LANES - 1) / DLAGP_LEVEL_LANES;
    const long lo = (long)vt * chunk < S ? (long)vt * chunk : S, hi = lo + chunk < S ? lo + chunk : S;
    double top = -__builtin_inf();
    double kept[BATCH];
    for (long i0 = lo; i0 < hi; i0 += BATCH) {
        double v[BATCH];
        double zs[BATCH][DLAGP_MAX_ABSORBERS_DEV];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const long i = i0 + u;
            const long ii = i < hi ? i : hi - 1;
            v[u] = in[ii] - a.log_s;
            if (level > 0)
#pragma unroll
                for (int j = 0; j < DLAGP_MAX_ABSORBERS_DEV; ++j)
                    if (j < kd) zs[u][j] = sample_z(a.src, b, ii, j, S, kd);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const long i = i0 + u;
            double value = v[u];
            if (level > 0) {
                bool close = false;
#pragma unroll
                for (int p = 1; p < DLAGP_MAX_ABSORBERS_DEV; ++p)
#pragma unroll
                    for (int q = 0; q < p; ++q)
                        if (p < kd) close |= fabs(zs[u][p] - zs[u][q]) < a.min_z_separation;
                if (close) value = __builtin_nan("");
            }
            kept[u] = value;
            if (i < hi) {
                col[i] = value;
                if (value == value) top = fmax(top, value);
            }
        }
    }
    const bool in_registers = hi - lo <= BATCH;  // the chunk was one batch: its values are still here
    double sum = 0.0, count = 0.0;
    for (long i0 = lo; i0 < hi; i0 += BATCH) {
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const long i = i0 + u;
            if (i >= hi) continue;
            const double value = in_registers ? kept[u] : col[i];
            double w = 0.0;
            if (value == value) {
                w = exp(value - top);
                count += 1.0;
            }
            sum += w;
            W[i] = w;
        }
    }
    for (int off = 32; off > 0; off >>= 1) count += __shfl_xor(count, off);
    x_m(a, b)[vt] = top;
    x_s(a, b)[vt] = sum;
    if ((threadIdx.x & 63) == 0) x_cnt(a, b)[vt >> 6] = count;
}

// LDS of phase B besides the cumulative array
struct ResampleShared {
    double scale[DLAGP_LEVEL_LANES];  // f_t = exp(m_t - M)
    double incl[DLAGP_LEVEL_LANES];   // inclusive scan of f_t s_t within the lane's wave
    double tot[16], offs[18], red[16];
};

// Phase B, a workgroup of `threads` threads (a multiple of 64, a divisor of 1024) that is number `group` of `groups` of
// its sightline: the level's maximum, the lanes' scales and scans, the evidence (group 0 writes it), the cumulative
// weights of ALL samples into `cum` (LDS; or, IN_PLACE, over the weights in memory -- one group only), then this
// group's share of the searches.
// LANES_PER_THREAD = 1024 / threads, a compile-time bound of the turns a thread takes over the virtual lanes: everything this
// phase reads -- the lanes' maxima and sums, the wave counts, the weights of the thread's chunks, its first uniforms -- has an
// address that is known when the kernel starts, so all of it is REQUESTED first and used later: four dependent trips to L2
// (1.5 us each on a workgroup that has nothing else to do) become one.
template <bool IN_PLACE, int LANES_PER_THREAD>
__device__ __forceinline__ void resample(const LevelArgs &a, long b, int tid, int threads, int group, int groups, double *cum_lds,
                                         ResampleShared &sh) {
    const long S = a.S;
    const int level = a.level;
    const double *xm = x_m(a, b), *xs = x_s(a, b);
    const int lane = tid & 63;
    const bool resamples = level + 1 < a.max_dlas;
    const double *W = a.weights + b * S;
    const long chunk = (S + DLAGP_LEVEL_LANES - 1) / DLAGP_LEVEL_LANES;
    constexpr int KEEP = 12;  // S <= 12 288: a chunk's reads are issued together, in front of everything else
    constexpr int BATCH = 5;  // searches per thread in flight: each is ~14 dependent reads of the cumulative array
    const bool prefetched = resamples && chunk <= KEEP;
    const long stride = (long)groups * threads;
    const long first = (long)group * threads + tid;
    const double *u = a.uniforms + (b * (a.max_dlas - 1) + level) * S;

    // ---- every read whose address does not depend on a result of this kernel
    double m[LANES_PER_THREAD], sums[LANES_PER_THREAD], w[LANES_PER_THREAD][KEEP], u_first[BATCH];
#pragma unroll
    for (int j = 0; j < LANES_PER_THREAD; ++j) {
        const int vl = tid + j * threads;
        m[j] = xm[vl];
        sums[j] = xs[vl];
    }
    const double my_count = tid < 16 ? x_cnt(a, b)[tid] : 0.0;
    if (prefetched) {
#pragma unroll
        for (int j = 0; j < LANES_PER_THREAD; ++j) {
            const long lo = (long)(tid + j * threads) * chunk, hi = lo + chunk < S ? lo + chunk : S;
#pragma unroll
            for (int k = 0; k < KEEP; ++k) w[j][k] = lo + k < hi ? W[lo + k] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const long i = first + (long)k * stride;
            u_first[k] = u[i < S ? i : S - 1];
        }
    }

    double top = -__builtin_inf();
#pragma unroll
    for (int j = 0; j < LANES_PER_THREAD; ++j) top = fmax(top, m[j]);
    top = wave_max(top);
    if (lane == 0) sh.red[tid >> 6] = top;
    if (tid < 16) sh.tot[tid] = my_count;  // parked here until the scans below replace it (read in between by thread 0 only)
    __syncthreads();
    top = sh.red[0];
    for (int wv = 1; wv < threads / 64; ++wv) top = fmax(top, sh.red[wv]);
    double count = 0.0;
    if (tid == 0)
        for (int wv = 0; wv < 16; ++wv) count += sh.tot[wv];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LANES_PER_THREAD; ++j) {  // the real wave holds one whole virtual wave per turn
        const int vl = tid + j * threads;
        const double f = m[j] > -__builtin_inf() ? exp(m[j] - top) : 0.0;
        double incl = f * sums[j];
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        sh.scale[vl] = f;
        sh.incl[vl] = incl;
        if (lane == 63) sh.tot[vl >> 6] = incl;
    }
    __syncthreads();
    if (tid == 0) {
        double run = 0.0;
        for (int wv = 0; wv < 16; ++wv) {
            sh.offs[wv] = run;
            run += sh.tot[wv];
        }
        sh.offs[16] = run;
        sh.offs[17] = count;
    }
    __syncthreads();
    const double last = sh.offs[16];
    count = sh.offs[17];
    const double evidence = count > 0.0 ? top + log(last / count) - level * a.log_s : __builtin_nan("");
    if (group == 0 && tid == 0) {
        a.evidences[b * a.max_dlas + level] = evidence;
        if (!(evidence == evidence)) a.alive[b] = 0;
    }
    if (!resamples || !(evidence == evidence)) return;

    double *cum = IN_PLACE ? a.weights + b * S : cum_lds;
#pragma unroll
    for (int j = 0; j < LANES_PER_THREAD; ++j) {
        const int vl = tid + j * threads;
        const long lo = (long)vl * chunk, hi = lo + chunk < S ? lo + chunk : S;
        if (lo >= S) break;
        const double f = sh.scale[vl];
        double run = sh.offs[vl >> 6] + ((vl & 63) ? sh.incl[vl - 1] : 0.0);
        if (prefetched) {
#pragma unroll
            for (int k = 0; k < KEEP; ++k)
                if (lo + k < hi) {
                    run += f * w[j][k];
                    cum[lo + k] = run;
                }
        } else {
            for (long i = lo; i < hi; ++i) {
                run += f * W[i];
                cum[i] = run;
            }
        }
    }
    __syncthreads();

    int *out = a.base + (b * (a.max_dlas - 1) + level) * S;
    const double margin = DLAGP_TIE_MARGIN * last;
    int near = 0;
    bool first_turn = true;
    for (long i0 = first; i0 < S; i0 += (long)BATCH * stride) {
        double x[BATCH];
        long left[BATCH], right[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const long i = i0 + (long)k * stride;
            x[k] = (first_turn && prefetched ? u_first[k] : u[i < S ? i : S - 1]) * last;
            left[k] = 0;
            right[k] = i < S ? S : 0;  // first index with cum[index] > x  (searchsorted side="right")
        }
        first_turn = false;
        bool busy = true;
        while (busy) {
            busy = false;
#pragma unroll
            for (int k = 0; k < BATCH; ++k)
                if (left[k] < right[k]) {
                    const long mid = (left[k] + right[k]) >> 1;
                    if (cum[mid] <= x[k]) left[k] = mid + 1;
                    else right[k] = mid;
                    busy = true;
                }
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const long i = i0 + (long)k * stride;
            if (i >= S) continue;
            const long at = left[k];
            out[i] = (int)(at < S ? at : S - 1);
            // the steps on either side of the answer: cum[at] (the first above x) and cum[at - 1] (the last not above)
            const bool above = at < S && cum[at] - x[k] <= margin;
            const bool below = at > 0 && x[k] - cum[at - 1] <= margin;
            near += (above || below) ? 1 : 0;
        }
    }
    if (a.ties) {
        for (int off = 32; off > 0; off >>= 1) near += __shfl_xor(near, off);
        if (lane == 0 && near) atomicAdd(a.ties + b * (a.max_dlas - 1) + level, near);
    }
}

// bookkeeping every form does first, by one thread of the launch's first kernel
__device__ __forceinline__ void begin(const LevelArgs &a, long b, bool first_block, int tid) {
    if (first_block && b == 0 && tid == 0 && *a.status) *a.sticky_status = 1;  // the next level's accumulate clears *status
    if (first_block && tid == 0 && a.ties && a.level + 1 < a.max_dlas) a.ties[b * (a.max_dlas - 1) + a.level] = 0;
}
}  // namespace lvl

// ---- the one-workgroup form
template <bool CUM_IN_LDS>
__global__ __launch_bounds__(DLAGP_LEVEL_THREADS) void level_finish_fused(LevelArgs a) {
    extern __shared__ double cum_lds[];
    __shared__ lvl::ResampleShared sh;
    const int tid = threadIdx.x;
    const long b = blockIdx.x;
    lvl::begin(a, b, true, tid);
    if (!a.alive[b]) return;  // an earlier level's evidence was NaN: the reference stopped there
    lvl::lane_weights<4>(a, b, tid);
    __syncthreads();
    lvl::resample<!CUM_IN_LDS, 1>(a, b, tid, DLAGP_LEVEL_THREADS, 0, 1, cum_lds, sh);
}

// ---- the spread form: grid (workgroups per sightline, sightlines)
__global__ __launch_bounds__(64) void level_weights(LevelArgs a) {
    const long b = blockIdx.y;
    lvl::begin(a, b, blockIdx.x == 0, threadIdx.x);
    if (!a.alive[b]) return;
    lvl::lane_weights<12>(a, b, (int)blockIdx.x * 64 + (int)threadIdx.x);  // grid.x == 16: one wave of virtual lanes per workgroup
}

__global__ __launch_bounds__(DLAGP_LEVEL_SEARCH_THREADS) void level_resample(LevelArgs a) {
    extern __shared__ double cum_lds[];
    __shared__ lvl::ResampleShared sh;
    const long b = blockIdx.y;
    if (!a.alive[b]) return;
    lvl::resample<false, DLAGP_LEVEL_LANES / DLAGP_LEVEL_SEARCH_THREADS>(a, b, threadIdx.x, DLAGP_LEVEL_SEARCH_THREADS, blockIdx.x,
                                                                         gridDim.x, cum_lds, sh);
}

// ---- two level finishes side by side (LevelArgsTwo): the same two kernels, the grid's rows dealt to two argument sets
__global__ __launch_bounds__(64) void level_weights_two(LevelArgsTwo p) {
    const bool second = (int)blockIdx.y >= p.nb_first;
    const LevelArgs &a = second ? p.second : p.first;
    const long b = (long)blockIdx.y - (second ? p.nb_first : 0);
    if (!second) lvl::begin(a, b, blockIdx.x == 0, threadIdx.x);
    if (!a.alive[b]) return;
    lvl::lane_weights<12>(a, b, (int)blockIdx.x * 64 + (int)threadIdx.x);
}

__global__ __launch_bounds__(DLAGP_LEVEL_SEARCH_THREADS) void level_resample_two(LevelArgsTwo p) {
    extern __shared__ double cum_lds[];
    __shared__ lvl::ResampleShared sh;
    const bool second = (int)blockIdx.y >= p.nb_first;
    const LevelArgs &a = second ? p.second : p.first;
    const long b = (long)blockIdx.y - (second ? p.nb_first : 0);
    if (!a.alive[b]) return;
    lvl::resample<false, DLAGP_LEVEL_LANES / DLAGP_LEVEL_SEARCH_THREADS>(a, b, threadIdx.x, DLAGP_LEVEL_SEARCH_THREADS, blockIdx.x,
                                                                         gridDim.x, cum_lds, sh);
}

// Start of an occupation sweep in one launch (reference dla_gp.py:112-119: NaN-initialised likelihoods and evidences,
// zeroed base indices; plus this library's flags): five fill launches before, 20 us of a one-spectrum call.
// `sample_ll2`, `evidences2`, `alive2`: the state of a second, one-level model started with the sweep (the paired one-absorber
// model of a model selection: its own likelihoods, evidences and alive flags; no sticky flag, no base indices) -- or NULL.
__global__ void sweep_init(double *__restrict__ sample_ll, long n_ll, double *__restrict__ evidences, long n_ev,
                           int *__restrict__ alive, long nb, int clear_sticky, int *__restrict__ base, long n_base,
                           double *__restrict__ sample_ll2, long n_ll2, double *__restrict__ evidences2, long n_ev2,
                           int *__restrict__ alive2) {
    const double nan = __builtin_nan("");
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_ll; i += stride) sample_ll[i] = nan;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_base; i += stride) base[i] = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_ll2; i += stride) sample_ll2[i] = nan;
    if (blockIdx.x == 0) {
        for (long i = threadIdx.x; i < n_ev; i += blockDim.x) evidences[i] = nan;
        for (long i = threadIdx.x; i < nb; i += blockDim.x) alive[i] = 1;
        if (threadIdx.x == 0 && clear_sticky) alive[nb] = 0;
        for (long i = threadIdx.x; i < n_ev2; i += blockDim.x) evidences2[i] = nan;
        if (alive2) {
            for (long i = threadIdx.x; i < nb; i += blockDim.x) alive2[i] = 1;
            if (threadIdx.x == 0) alive2[nb] = 0;
        }
    }
}

// grid (blocks, segments): see CopyTable (dlagp_device.h).  ONE-WAVE workgroups, like every kernel that runs beside other
// batches' sweeps (host_shared.h, DLAGP_SETUP_THREADS): a workgroup of four waves finds four free wave slots on one compute
// unit only when a sweep's launch drains -- with 256 threads this kernel made a catalogue run's kernel stream run dry at 13 %
// of the instants (2060 instead of 2520 spectra/s).
__global__ __launch_bounds__(DLAGP_SETUP_THREADS) void copy_segments(CopyTable t, unsigned char *__restrict__ pack) {
    const CopySegment s = t.seg[blockIdx.y];
    const size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if (((s.device_ptr | s.bytes) & 15) == 0) {  // the pack offset is a multiple of 256
        uint4 *dev = reinterpret_cast<uint4 *>(s.device_ptr);
        uint4 *pk = reinterpret_cast<uint4 *>(pack + s.offset);
        for (size_t i = first; i < s.bytes / 16; i += stride)
            if (t.to_pack) pk[i] = dev[i];
            else dev[i] = pk[i];
    } else {
        unsigned int *dev = reinterpret_cast<unsigned int *>(s.device_ptr);
        unsigned int *pk = reinterpret_cast<unsigned int *>(pack + s.offset);
        for (size_t i = first; i < s.bytes / 4; i += stride)
            if (t.to_pack) pk[i] = dev[i];
            else dev[i] = pk[i];
    }
}

__global__ void fill_f64(double *__restrict__ dst, long n, double value) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = value;
}

__global__ void fill_i32(int *__restrict__ dst, long n, int value) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = value;
}

}  // namespace dlagp
