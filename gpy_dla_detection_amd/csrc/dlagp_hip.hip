// libdlagp_hip.so -- host side of the C ABI declared in include/dlagp.h, and the single
// translation unit that instantiates every kernel (one HIP module per device).
//
// Build: make -C gpy_dla_detection_amd/csrc   (hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dlagp.h"
#include "dlagp_device.h"
#include "kernels_common.h"
#include "kernels_generic.h"
#include "voigt_device.h"

using namespace dlagp;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t err__ = (expr);                                                                 \
        if (err__ != hipSuccess)                                                                   \
            return fail(DLAGP_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__),  \
                        __FILE__, __LINE__);                                                       \
    } while (0)

// ------------------------------------------------------------------------------------------------
// options and timing
// ------------------------------------------------------------------------------------------------
static int64_t g_opt_pixel_splits = 0;
static int64_t g_opt_path = 0;
static int64_t g_opt_timing = 0;

struct EventPair {
    hipEvent_t start, stop;
};
static std::mutex g_timing_mutex;
static std::vector<EventPair> g_timing_events;

// ------------------------------------------------------------------------------------------------
// per-device one-time setup: Voigt line constants into constant memory
// ------------------------------------------------------------------------------------------------
static std::mutex g_init_mutex;
static std::vector<int> g_device_ready;

static int ensure_device(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(DLAGP_EDEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= count) return fail(DLAGP_EINVAL, "device %d out of range [0, %d)", device, count);
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_init_mutex);
    if ((int)g_device_ready.size() < count) g_device_ready.resize(count, 0);
    if (g_device_ready[device]) return DLAGP_OK;

    LineConstants lc;
    const double sigma = DLAGP_SIGMA_CGS;
    const double sqrt2_sigma = std::sqrt(2.0) * sigma;
    const double sqrt_2pi_sigma = std::sqrt(2.0 * M_PI) * sigma;
    for (int l = 0; l < DLAGP_MAX_LINES; ++l) {
        lc.lambda_cm[l] = DLAGP_TRANSITION_WAVELENGTHS_CM[l];
        lc.y[l] = DLAGP_LORENTZ_WIDTHS[l] / sqrt2_sigma;
        lc.k_core[l] = DLAGP_LEADING_CONSTANTS[l] / sqrt_2pi_sigma;
        lc.k_wing[l] = lc.k_core[l] * lc.y[l] / std::sqrt(M_PI);
    }
    lc.inv_sqrt2_sigma = 1.0 / sqrt2_sigma;
    lc.c = DLAGP_SPEED_OF_LIGHT_CGS;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(DLAGP_LINES), &lc, sizeof(lc)));
    g_device_ready[device] = 1;
    return DLAGP_OK;
}

// ------------------------------------------------------------------------------------------------
// device buffers
// ------------------------------------------------------------------------------------------------
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t want) {
        if (want <= bytes) return DLAGP_OK;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
        HIP_TRY(hipMalloc(&ptr, want));
        bytes = want;
        return DLAGP_OK;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    template <typename T>
    T *as() const {
        return static_cast<T *>(ptr);
    }
};

// scratch shared by the launches of one context / batch
struct Workspace {
    DeviceBuffer part;    // partial rows [nsplit][rows][row_width]
    DeviceBuffer status;  // int flag raised by finalize on a non-PD pivot
    DeviceBuffer z, nhi, out;  // staging for the host-pointer entry points
    void release() {
        part.release();
        status.release();
        z.release();
        nhi.release();
        out.release();
    }
};

struct dlagp_ctx {
    int device = 0;
    SightlineDev sl{};         // host copy of the descriptor (device pointers inside)
    DeviceBuffer d_sl;         // the same descriptor in HBM (table of one)
    DeviceBuffer y, mu, omega2, v, pos, lam, P;
    Workspace ws;
};

struct dlagp_batch {
    int device = 0;
    std::vector<dlagp_ctx *> ctxs;
    DeviceBuffer d_table;  // SightlineDev[nb]
    int max_entries = 0, max_npad = 0;
    bool uniform_rank = true;
    int k = 0, row_width = 0;
    Workspace ws;
};

// ------------------------------------------------------------------------------------------------
// launch sequence common to single-sightline and batch evaluation
// ------------------------------------------------------------------------------------------------
struct LaunchShape {
    const SightlineDev *d_table;
    int nb;
    int k;          // common rank
    int row_width;  // common partial-row width
    int max_entries, max_npad;
};

static int launch_sweep(const LaunchShape &shape, Workspace &ws, int64_t S, int kd, const double *d_z,
                        const double *d_nhi, double *d_out, hipStream_t stream) {
    if (S <= 0) return DLAGP_OK;
    const long rows = (long)shape.nb * S;
    const int nsplit = 1;
    int rc = ws.part.reserve((size_t)nsplit * rows * shape.row_width * sizeof(double));
    if (rc) return rc;
    rc = ws.status.reserve(sizeof(int));
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(ws.status.ptr, 0, sizeof(int), stream));

    EventPair ev{};
    const bool timing = g_opt_timing != 0;
    if (timing) {
        HIP_TRY(hipEventCreate(&ev.start));
        HIP_TRY(hipEventCreate(&ev.stop));
        HIP_TRY(hipEventRecord(ev.start, stream));
    }

    {
        // w[] (n_pad values) reuses the raw-profile buffer: size it for both
        const int lds_entries = std::max(shape.max_entries, shape.max_npad);
        const int lds_pixels = shape.max_npad;
        const size_t lds_bytes = (size_t)(lds_entries + lds_pixels + DLAGP_MAX_LINES + 1 + 8) * sizeof(double);
        if (lds_bytes > 160 * 1024)
            return fail(DLAGP_EINVAL, "sightline too long for the generic kernel: %zu B of LDS needed", lds_bytes);
        if (lds_bytes > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(accumulate_generic),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        dim3 grid((unsigned)S, (unsigned)shape.nb);
        hipLaunchKernelGGL(accumulate_generic, grid, dim3(DLAGP_GENERIC_THREADS), lds_bytes, stream,
                           shape.d_table, (long)S, kd, d_z, d_nhi, ws.part.as<double>(), lds_entries,
                           lds_pixels);
        HIP_TRY(hipGetLastError());
    }

    if (timing) {
        HIP_TRY(hipEventRecord(ev.stop, stream));
        std::lock_guard<std::mutex> lock(g_timing_mutex);
        g_timing_events.push_back(ev);
    }

    const int threads = DLAGP_FINALIZE_THREADS;
    const unsigned blocks = (unsigned)((rows + threads - 1) / threads);
    if (shape.k == 20) {
        hipLaunchKernelGGL(finalize_samples<20>, dim3(blocks), dim3(threads), 0, stream,
                           ws.part.as<double>(), nsplit, rows, (long)S, shape.d_table, d_out,
                           ws.status.as<int>());
    } else {
        hipLaunchKernelGGL(finalize_samples<0>, dim3(blocks), dim3(threads), 0, stream,
                           ws.part.as<double>(), nsplit, rows, (long)S, shape.d_table, d_out,
                           ws.status.as<int>());
    }
    HIP_TRY(hipGetLastError());
    return DLAGP_OK;
}

// host-pointer wrapper: stage, launch, copy back, report non-PD as DLAGP_ELINALG
static int sweep_from_host(const LaunchShape &shape, Workspace &ws, int64_t S, int kd, const double *z,
                           const double *nhi, double *out) {
    const long rows = (long)shape.nb * S;
    const size_t nsamp = (size_t)rows * (kd > 0 ? kd : 1);
    int rc;
    if ((rc = ws.z.reserve(nsamp * sizeof(double)))) return rc;
    if ((rc = ws.nhi.reserve(nsamp * sizeof(double)))) return rc;
    if ((rc = ws.out.reserve((size_t)rows * sizeof(double)))) return rc;
    if (kd > 0) {
        HIP_TRY(hipMemcpy(ws.z.ptr, z, nsamp * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ws.nhi.ptr, nhi, nsamp * sizeof(double), hipMemcpyHostToDevice));
    }
    rc = launch_sweep(shape, ws, S, kd, ws.z.as<double>(), ws.nhi.as<double>(), ws.out.as<double>(), nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, ws.out.ptr, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost));
    int status = 0;
    HIP_TRY(hipMemcpy(&status, ws.status.ptr, sizeof(int), hipMemcpyDeviceToHost));
    if (status) return fail(DLAGP_ELINALG, "Matrix is not positive definite (I + M' D^-1 M)");
    return DLAGP_OK;
}

static LaunchShape shape_of(dlagp_ctx *ctx) {
    return LaunchShape{ctx->d_sl.as<SightlineDev>(), 1, ctx->sl.k, ctx->sl.row_width, ctx->sl.num_entries,
                       ctx->sl.n_pad};
}

static LaunchShape shape_of(dlagp_batch *b) {
    return LaunchShape{b->d_table.as<SightlineDev>(), (int)b->ctxs.size(), b->k, b->row_width, b->max_entries,
                       b->max_npad};
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int dlagp_abi_version(void) { return 1; }

int dlagp_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

const char *dlagp_last_error(void) { return g_last_error.c_str(); }

int dlagp_create(dlagp_ctx **out, int n, int n_u, int k, const double *y, const double *mu, const double *M,
                 ptrdiff_t m_row_stride, ptrdiff_t m_col_stride, const double *omega2, const double *v,
                 const double *wavelengths, const uint8_t *keep, int num_lines, int broadening, int device) {
    if (!out) return fail(DLAGP_EINVAL, "out is NULL");
    *out = nullptr;
    if (!y || !mu || !M || !omega2 || !v || !wavelengths || !keep) return fail(DLAGP_EINVAL, "NULL input array");
    if (n <= 0 || n_u < n) return fail(DLAGP_EINVAL, "need 0 < n <= n_u (n=%d, n_u=%d)", n, n_u);
    if (k < 1 || k > DLAGP_MAX_RANK) return fail(DLAGP_EINVAL, "rank k=%d outside [1, %d]", k, DLAGP_MAX_RANK);
    if (num_lines < 1 || num_lines > DLAGP_MAX_LINES)
        return fail(DLAGP_EINVAL, "num_lines=%d outside [1, %d]", num_lines, DLAGP_MAX_LINES);
    int kept = 0;
    for (int j = 0; j < n_u; ++j) kept += keep[j] != 0;
    if (kept != n)
        return fail(DLAGP_EINVAL, "keep flags select %d pixels but n=%d (len(absorption) == len(this_mu))", kept, n);
    int rc = ensure_device(device);
    if (rc) return rc;

    dlagp_ctx *ctx = new dlagp_ctx();
    ctx->device = device;
    SightlineDev &sl = ctx->sl;
    sl.n = n;
    sl.n_pad = (n + DLAGP_STEP_PIXELS - 1) / DLAGP_STEP_PIXELS * DLAGP_STEP_PIXELS;
    sl.nsteps = sl.n_pad / DLAGP_STEP_PIXELS;
    sl.k = k;
    sl.ntaps = broadening ? DLAGP_NUM_TAPS : 1;
    sl.num_entries = broadening ? n_u + DLAGP_NUM_TAPS - 1 : n_u;
    sl.num_lines = num_lines;
    sl.ntw = (k * (k + 1) / 2 + DLAGP_TILE_COLS - 1) / DLAGP_TILE_COLS;
    sl.ntq = (k + DLAGP_TILE_COLS - 1) / DLAGP_TILE_COLS;
    sl.row_width = (sl.ntw + sl.ntq) * DLAGP_TILE_COLS + DLAGP_ROW_EXTRA;
    for (int t = 0; t < DLAGP_NUM_TAPS; ++t) sl.taps[t] = broadening ? DLAGP_INSTRUMENT_PROFILE[t] : (t == 0 ? 1.0 : 0.0);

    // padded host images
    std::vector<double> hy(sl.n_pad, 0.0), hmu(sl.n_pad, 0.0), ho(sl.n_pad, 0.0), hv(sl.n_pad, 1.0);
    std::vector<int> hpos(sl.n_pad, 0);
    std::vector<double> hM((size_t)n * k);
    for (int i = 0; i < n; ++i) {
        hy[i] = y[i];
        hmu[i] = mu[i];
        ho[i] = omega2[i];
        hv[i] = v[i];
        for (int c = 0; c < k; ++c) hM[(size_t)i * k + c] = M[i * m_row_stride + c * m_col_stride];
    }
    // window start of kept pixel i = index of its unmasked pixel: with broadening the padded grid
    // has 3 extra entries in front, so the 7-tap "valid" window of unmasked pixel j is j .. j+6
    for (int j = 0, i = 0; j < n_u; ++j)
        if (keep[j]) hpos[i++] = j;

    auto upload = [&](DeviceBuffer &buf, const void *src, size_t bytes) -> int {
        int r = buf.reserve(bytes);
        if (r) return r;
        HIP_TRY(hipMemcpy(buf.ptr, src, bytes, hipMemcpyHostToDevice));
        return DLAGP_OK;
    };
    DeviceBuffer dM;
    const int nt = sl.ntw + sl.ntq;
    const size_t p_bytes = (size_t)sl.nsteps * nt * DLAGP_TILE_DOUBLES * sizeof(double);
    rc = upload(ctx->y, hy.data(), hy.size() * sizeof(double));
    if (!rc) rc = upload(ctx->mu, hmu.data(), hmu.size() * sizeof(double));
    if (!rc) rc = upload(ctx->omega2, ho.data(), ho.size() * sizeof(double));
    if (!rc) rc = upload(ctx->v, hv.data(), hv.size() * sizeof(double));
    if (!rc) rc = upload(ctx->pos, hpos.data(), hpos.size() * sizeof(int));
    if (!rc) rc = upload(ctx->lam, wavelengths, (size_t)sl.num_entries * sizeof(double));
    if (!rc) rc = upload(dM, hM.data(), hM.size() * sizeof(double));
    if (!rc) rc = ctx->P.reserve(p_bytes);
    if (rc) {
        dM.release();
        dlagp_destroy(ctx);
        return rc;
    }
    sl.y = ctx->y.as<double>();
    sl.mu = ctx->mu.as<double>();
    sl.omega2 = ctx->omega2.as<double>();
    sl.v = ctx->v.as<double>();
    sl.pos = ctx->pos.as<int>();
    sl.lam = ctx->lam.as<double>();
    sl.P = ctx->P.as<double>();

    hipLaunchKernelGGL(build_shared_operand, dim3(1024), dim3(256), 0, nullptr, dM.as<double>(), n, sl.n_pad, k,
                       sl.ntw, sl.ntq, ctx->P.as<double>());
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipDeviceSynchronize();
    dM.release();
    if (err != hipSuccess) {
        dlagp_destroy(ctx);
        return fail(DLAGP_EDEVICE, "build_shared_operand failed: %s", hipGetErrorString(err));
    }
    rc = upload(ctx->d_sl, &sl, sizeof(sl));
    if (rc) {
        dlagp_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return DLAGP_OK;
}

void dlagp_destroy(dlagp_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    ctx->d_sl.release();
    ctx->y.release();
    ctx->mu.release();
    ctx->omega2.release();
    ctx->v.release();
    ctx->pos.release();
    ctx->lam.release();
    ctx->P.release();
    ctx->ws.release();
    delete ctx;
}

int dlagp_ctx_n(const dlagp_ctx *ctx) { return ctx ? ctx->sl.n : DLAGP_EINVAL; }
int dlagp_ctx_k(const dlagp_ctx *ctx) { return ctx ? ctx->sl.k : DLAGP_EINVAL; }
int dlagp_ctx_device(const dlagp_ctx *ctx) { return ctx ? ctx->device : DLAGP_EINVAL; }

static int check_eval_args(const void *ctx, int64_t S, int kd, const void *z, const void *nhi, const void *out) {
    if (!ctx) return fail(DLAGP_EINVAL, "context is NULL");
    if (S < 0) return fail(DLAGP_EINVAL, "S=%lld is negative", (long long)S);
    if (kd < 1 || kd > DLAGP_MAX_ABSORBERS)
        return fail(DLAGP_EINVAL, "kd=%d outside [1, %d]", kd, DLAGP_MAX_ABSORBERS);
    if (S > 0 && (!z || !nhi || !out)) return fail(DLAGP_EINVAL, "NULL sample or output array");
    return DLAGP_OK;
}

int dlagp_eval(dlagp_ctx *ctx, int64_t S, int kd, const double *z, const double *nhi, double *out) {
    int rc = check_eval_args(ctx, S, kd, z, nhi, out);
    if (rc) return rc;
    if (S == 0) return DLAGP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    return sweep_from_host(shape_of(ctx), ctx->ws, S, kd, z, nhi, out);
}

int dlagp_null(dlagp_ctx *ctx, double *out) {
    if (!ctx || !out) return fail(DLAGP_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    return sweep_from_host(shape_of(ctx), ctx->ws, 1, 0, nullptr, nullptr, out);
}

int dlagp_mvnpdf(int n, int k, const double *y, const double *mu, const double *M, ptrdiff_t m_row_stride,
                 ptrdiff_t m_col_stride, const double *d, double *out, int device) {
    if (!y || !mu || !M || !d || !out) return fail(DLAGP_EINVAL, "NULL argument");
    if (n <= 0) return fail(DLAGP_EINVAL, "n=%d must be positive", n);
    // a = 1, omega2 = 0, v = d: the null-model density with diagonal d
    std::vector<double> zeros(n, 0.0), lam(n);
    std::vector<uint8_t> keep(n, 1);
    for (int i = 0; i < n; ++i) lam[i] = 4000.0 + i;
    dlagp_ctx *ctx = nullptr;
    int rc = dlagp_create(&ctx, n, n, k, y, mu, M, m_row_stride, m_col_stride, zeros.data(), d, lam.data(),
                          keep.data(), 1, 0, device);
    if (rc) return rc;
    rc = dlagp_null(ctx, out);
    dlagp_destroy(ctx);
    return rc;
}

int dlagp_absorption(dlagp_ctx *ctx, int kd, const double *z, const double *nhi, double *out) {
    int rc = check_eval_args(ctx, 1, kd, z, nhi, out);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    Workspace &ws = ctx->ws;
    if ((rc = ws.z.reserve(kd * sizeof(double)))) return rc;
    if ((rc = ws.nhi.reserve(kd * sizeof(double)))) return rc;
    if ((rc = ws.out.reserve((size_t)ctx->sl.n_pad * sizeof(double)))) return rc;
    HIP_TRY(hipMemcpy(ws.z.ptr, z, kd * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ws.nhi.ptr, nhi, kd * sizeof(double), hipMemcpyHostToDevice));
    const int lds_entries = std::max(ctx->sl.num_entries, ctx->sl.n_pad);
    const size_t lds_bytes = (size_t)(lds_entries + ctx->sl.n_pad + DLAGP_MAX_LINES + 1 + 8) * sizeof(double);
    if (lds_bytes > 160 * 1024) return fail(DLAGP_EINVAL, "sightline too long: %zu B of LDS needed", lds_bytes);
    if (lds_bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(absorption_generic),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(absorption_generic, dim3(1), dim3(DLAGP_GENERIC_THREADS), lds_bytes, nullptr,
                       ctx->d_sl.as<SightlineDev>(), kd, ws.z.as<double>(), ws.nhi.as<double>(), ws.out.as<double>(),
                       lds_entries, ctx->sl.n_pad);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, ws.out.ptr, (size_t)ctx->sl.n * sizeof(double), hipMemcpyDeviceToHost));
    return DLAGP_OK;
}

int dlagp_eval_device(dlagp_ctx *ctx, int64_t S, int kd, const double *d_z, const double *d_nhi, double *d_out,
                      void *stream) {
    int rc = check_eval_args(ctx, S, kd, d_z, d_nhi, d_out);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    return launch_sweep(shape_of(ctx), ctx->ws, S, kd, d_z, d_nhi, d_out, static_cast<hipStream_t>(stream));
}

int dlagp_batch_create(dlagp_batch **out, dlagp_ctx *const *ctxs, int nb) {
    if (!out) return fail(DLAGP_EINVAL, "out is NULL");
    *out = nullptr;
    if (!ctxs || nb <= 0) return fail(DLAGP_EINVAL, "empty batch");
    for (int b = 0; b < nb; ++b)
        if (!ctxs[b]) return fail(DLAGP_EINVAL, "context %d is NULL", b);
    dlagp_batch *batch = new dlagp_batch();
    batch->device = ctxs[0]->device;
    batch->k = ctxs[0]->sl.k;
    batch->row_width = ctxs[0]->sl.row_width;
    std::vector<SightlineDev> table(nb);
    for (int b = 0; b < nb; ++b) {
        if (ctxs[b]->device != batch->device) {
            delete batch;
            return fail(DLAGP_EINVAL, "context %d lives on device %d, batch on %d", b, ctxs[b]->device, ctxs[0]->device);
        }
        if (ctxs[b]->sl.k != batch->k) {
            delete batch;
            return fail(DLAGP_EINVAL, "context %d has rank %d, batch rank is %d", b, ctxs[b]->sl.k, batch->k);
        }
        batch->ctxs.push_back(ctxs[b]);
        table[b] = ctxs[b]->sl;
        batch->max_entries = std::max(batch->max_entries, ctxs[b]->sl.num_entries);
        batch->max_npad = std::max(batch->max_npad, ctxs[b]->sl.n_pad);
    }
    if (hipSetDevice(batch->device) != hipSuccess) {
        delete batch;
        return fail(DLAGP_EDEVICE, "hipSetDevice(%d) failed", batch->device);
    }
    int rc = batch->d_table.reserve(table.size() * sizeof(SightlineDev));
    if (!rc && hipMemcpy(batch->d_table.ptr, table.data(), table.size() * sizeof(SightlineDev),
                         hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(DLAGP_EDEVICE, "uploading the sightline table failed");
    if (rc) {
        dlagp_batch_destroy(batch);
        return rc;
    }
    *out = batch;
    return DLAGP_OK;
}

void dlagp_batch_destroy(dlagp_batch *batch) {
    if (!batch) return;
    (void)hipSetDevice(batch->device);
    batch->d_table.release();
    batch->ws.release();
    delete batch;
}

int dlagp_batch_eval_device(dlagp_batch *batch, int64_t S, int kd, const double *d_z, const double *d_nhi,
                            double *d_out, void *stream) {
    int rc = check_eval_args(batch, S, kd, d_z, d_nhi, d_out);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(batch->device));
    return launch_sweep(shape_of(batch), batch->ws, S, kd, d_z, d_nhi, d_out, static_cast<hipStream_t>(stream));
}

int dlagp_batch_eval(dlagp_batch *batch, int64_t S, int kd, const double *z, const double *nhi, double *out) {
    int rc = check_eval_args(batch, S, kd, z, nhi, out);
    if (rc) return rc;
    if (S == 0) return DLAGP_OK;
    HIP_TRY(hipSetDevice(batch->device));
    return sweep_from_host(shape_of(batch), batch->ws, S, kd, z, nhi, out);
}

int dlagp_set_option(const char *name, int64_t value) {
    if (!name) return fail(DLAGP_EINVAL, "option name is NULL");
    if (!strcmp(name, "pixel_splits")) g_opt_pixel_splits = value;
    else if (!strcmp(name, "path")) g_opt_path = value;
    else if (!strcmp(name, "timing")) g_opt_timing = value;
    else return fail(DLAGP_EINVAL, "unknown option '%s'", name);
    return DLAGP_OK;
}

int dlagp_timing_read(double *total_ms, int64_t *launches, int reset) {
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    double total = 0.0;
    for (const EventPair &ev : g_timing_events) {
        HIP_TRY(hipEventSynchronize(ev.stop));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.start, ev.stop));
        total += ms;
    }
    if (total_ms) *total_ms = total;
    if (launches) *launches = (int64_t)g_timing_events.size();
    if (reset) {
        for (const EventPair &ev : g_timing_events) {
            (void)hipEventDestroy(ev.start);
            (void)hipEventDestroy(ev.stop);
        }
        g_timing_events.clear();
    }
    return DLAGP_OK;
}

}  // extern "C"
