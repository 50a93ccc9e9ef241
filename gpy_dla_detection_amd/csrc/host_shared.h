// Constants and plain structs that BOTH sides of the library use: the kernels (kernels_*.h, device code) and the host
// runtime (dlagp_hip.hip) -- launch geometry, buffer sizes, launch flags, the learned-model descriptor.  No HIP header and no
// device code in here: the host-only sanitizer build of the runtime (runtime_replay.cpp over fake_device.h) includes this
// file and plan_host.h, and nothing of the kernels.
#pragma once
#include "dlagp_device.h"
#include "faddeeva_table.h"  // DLAGP_WING_COEF_INIT, DLAGP_EXP2_COEF_INIT: the host folds them into LineConstants
#include "plan_host.h"       // PassDesc, WorkItem, DLAGP_MFMA_GROUP, chunk / pass / tile sizes

// ---- MFMA path (kernels_mfma.h) ----
#define DLAGP_MFMA_RANK 20
#define DLAGP_MFMA_THREADS 64
#define DLAGP_MFMA_BIG_TILES 14
#define DLAGP_MFMA_W_TILES 13      // full tiles weighted by w; tile 13 is weighted by q
#define DLAGP_MFMA_PAIRS 8
#define DLAGP_MFMA_GROUP_W_COLS 210  // columns of vech(M_p M_p') one column group contracts (ranks above 20: several groups)
#define DLAGP_MFMA_GROUP_Q_COLS 20   // ... and of M_p
#define DLAGP_AUX_FIELDS 8       // per-pixel record: y, mu, omega2, v, window start, 3 spare
#define DLAGP_MFMA_ROW_WIDTH ((14 + 2) * DLAGP_TILE_COLS + DLAGP_ROW_EXTRA)  // 264: row_width of a rank-20 sightline (ntw 14, ntq 2)
#define DLAGP_LAUNCH_PIECES_MASK 0xff       // launch_flags, low byte: pieces per block of the launch plan (the partial sums'
                                           // stride between column groups)
#define DLAGP_LAUNCH_KEEP_STATUS 0x10000   // launch_flags: do not clear the non-PD flag (see accumulate_mfma)
#define DLAGP_LAUNCH_GENERAL_ROWS 0x20000  // cached levels: 16 row reads per pass even where one would do (option shared_rows = 0)

// ---- generic path (kernels_generic.h) ----
#define DLAGP_GENERIC_THREADS 256

// ---- finalisation, evidence, level_finish, interpolation (kernels_common.h) ----
#define DLAGP_FINALIZE_THREADS 64
#define DLAGP_FINW_SAMPLES 2
constexpr int finalize_wave_lds_doubles(int k) { return k * (k + 1) / 2 + k; }  // constexpr: callable from host and device code
#define DLAGP_FIN_K 20
#define DLAGP_FIN_COLS (DLAGP_FIN_K * (DLAGP_FIN_K + 1) / 2)
#define DLAGP_FIN_SAMPLES 16   // samples per workgroup: small, so that every CU pulls partial sums
#define DLAGP_FIN_THREADS 256
#define DLAGP_FIN_STAGED (DLAGP_FIN_COLS + DLAGP_FIN_K + 3)  // vech(B - I), u, 3 scalars
#define DLAGP_FIN_LDS_STRIDE 17  // doubles per staged column in LDS: odd, so that columns two apart fall on different banks
#define DLAGP_FINALIZE_LDS_DOUBLES (DLAGP_FIN_STAGED * DLAGP_FIN_LDS_STRIDE)
#ifndef DLAGP_FIN_ROW16_MAX_BLOCKS
#define DLAGP_FIN_ROW16_MAX_BLOCKS (1L << 40)
#endif
#define DLAGP_EVIDENCE_THREADS 1024
#define DLAGP_LEVEL_THREADS 1024
#define DLAGP_LEVEL_SEARCH_THREADS 256  // spread form of the level finish (kernels_common.h): threads per workgroup of level_resample
// Set-up kernels run as ONE-WAVE workgroups (round 5).  They are launched beside other batches' sweeps, whose accumulate
// waves hold every register of a SIMD and 19.5 KB of LDS each and are replaced the moment they end: a workgroup of four waves
// finds four free slots on ONE CU only when a sweep's launch drains -- interp_model averaged 263 us of wall time in a
// catalogue run where it computes for 5 (profiles/r05_experiments.txt item 9) -- while a single wave fits whichever slot
// frees next.
#define DLAGP_INTERP_PIXELS 2
#define DLAGP_INTERP_THREADS (32 * DLAGP_INTERP_PIXELS)
#define DLAGP_SETUP_THREADS 64

// the learned GP model on its rest-frame grid, as interp_model takes it (by value)
namespace dlagp {
struct ModelDev {
    int n_grid, k, num_forest_lines;
    const double *grid;       // [n_grid] rest wavelengths, increasing
    const double *mu;         // [n_grid]
    const double *log_omega;  // [n_grid]
    const double *M;          // [k][n_grid]: one contiguous column per eigenvector
    double prev_beta, beta, c_0;
    double lines[DLAGP_MAX_LINES];        // series wavelengths, Angstrom
    double scale_prev[DLAGP_MAX_LINES];   // prev_tau_0 f_l/f_0 lambda_l/lambda_0
    double scale_noise[DLAGP_MAX_LINES];  // tau_0        f_l/f_0 lambda_l/lambda_0
};
}  // namespace dlagp
