// libdlagp_hip.so -- host side of the C ABI declared in include/dlagp.h, and the single
// translation unit that instantiates every kernel (one HIP module per device).
//
// Build: make -C gpy_dla_detection_amd/csrc   (hipcc --offload-arch=gfx950)
#ifndef DLAGP_FAKE_DEVICE
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#else
// host-only sanitizer build of THIS file's host side (runtime_replay.cpp, `make sanitize`): a stand-in for the HIP runtime
// API with the stream / allocation semantics the bookkeeping below relies on; the kernels do not exist there.  Test
// infrastructure -- libdlagp_hip.so is never built this way.
#include "fake_device.h"
#define DLAGP_DEVICE_CHECK_IDLE(p) ((void)fake::check_idle((p), "DeviceBuffer::release"))
#endif
#ifndef DLAGP_DEVICE_CHECK_IDLE
#define DLAGP_DEVICE_CHECK_IDLE(p) ((void)0)
#endif

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/dlagp.h"
#include "dlagp_device.h"
#include "host_shared.h"
#include "prepare_host.h"
#ifndef DLAGP_FAKE_DEVICE
#include "kernels_common.h"
#include "kernels_generic.h"
#include "kernels_mfma.h"
#include "voigt_device.h"
#else
#include "fake_kernels.h"
#endif

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

// Exception barrier of the C ABI (include/dlagp.h: "Nothing aborts the process").  The host side uses std::vector,
// std::map, std::string and new; an exception that left an extern "C" function would unwind into the caller's frames --
// ctypes, cgo, JNI -- and end in std::terminate.  Every exported function is a function-try-block whose handler turns
// whatever was thrown into a status code and a message (DLAGP_CATCH_INT; the destroy functions swallow: DLAGP_CATCH_VOID).
static int exception_status(const char *where) noexcept {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        try {
            return fail(DLAGP_ENOMEM, "%s: out of host memory (std::bad_alloc)", where);
        } catch (...) {  // the message itself could not be stored
            return DLAGP_ENOMEM;
        }
    } catch (const std::length_error &e) {
        try {
            return fail(DLAGP_ENOMEM, "%s: a size exceeds what the host can address (%s)", where, e.what());
        } catch (...) {
            return DLAGP_ENOMEM;
        }
    } catch (const std::exception &e) {
        try {
            return fail(DLAGP_EINVAL, "%s: unexpected C++ exception: %s", where, e.what());
        } catch (...) {
            return DLAGP_EINVAL;
        }
    } catch (...) {
        try {
            return fail(DLAGP_EINVAL, "%s: unexpected C++ exception of unknown type", where);
        } catch (...) {
            return DLAGP_EINVAL;
        }
    }
}
#define DLAGP_CATCH_INT(name) catch (...) { return exception_status(name); }
#define DLAGP_CATCH_VOID(name) catch (...) { (void)exception_status(name); }

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
// Options and last-launch diagnostics are process-wide and may be read by a worker thread (context creation,
// batch bookkeeping) while another thread sets them: relaxed atomics, each read a plain int64_t.
struct Option {
    std::atomic<int64_t> v;
    constexpr Option(int64_t init) : v(init) {}
    operator int64_t() const { return v.load(std::memory_order_relaxed); }
    Option &operator=(int64_t x) {
        v.store(x, std::memory_order_relaxed);
        return *this;
    }
};
static Option g_opt_pixel_splits = 0;
static Option g_opt_path = 0;
static Option g_opt_timing = 0;
static Option g_opt_debug = 0;  // bit 1: stamped instantiation of the MFMA kernel (diagnostics)
static Option g_opt_core_patch = 1;     // MFMA kernel: line cores repaired outside the sample loop when eligible
static Option g_opt_fuse_finalize = 1;  // single-piece plans: the accumulate waves finish their own blocks (see fused_eligible)
static Option g_opt_pair_models = 1;    // the sub-DLA sweep rides in level 1's launch of the DLA sweep (see pair_eligible)

// The options that decide a call's launch SHAPES are read once per exported call and the call works from that snapshot:
// another thread's dlagp_set_option in the middle of a sweep (tools and tests do that) can then neither make the levels of
// one sweep disagree about its shape nor leave a two-phase call half started.  Nested scopes keep the outermost snapshot.
struct CallOptions {
    int64_t path, debug, core_patch, fuse_finalize, pair_models;
};
static CallOptions read_call_options() { return {g_opt_path, g_opt_debug, g_opt_core_patch, g_opt_fuse_finalize, g_opt_pair_models}; }
static thread_local const CallOptions *t_call_options = nullptr;
struct CallOptionsScope {
    CallOptions snapshot;
    bool mine;
    CallOptionsScope() : snapshot(read_call_options()), mine(t_call_options == nullptr) {
        if (mine) t_call_options = &snapshot;
    }
    ~CallOptionsScope() {
        if (mine) t_call_options = nullptr;
    }
    CallOptionsScope(const CallOptionsScope &) = delete;
    CallOptionsScope &operator=(const CallOptionsScope &) = delete;
};
static CallOptions call_options() { return t_call_options ? *t_call_options : read_call_options(); }
static Option g_opt_profile_cache = 1;  // multi-level sweeps keep level-1 profiles in HBM (0 = recompute)
// never more than this for the cache (MB), and never more than half of the free HBM
static Option g_opt_profile_cache_max_mb = 196608;
static Option g_opt_plan_item_cost = 20;   // fixed cost of a work item in the planner's model, tenths of a chunk
static Option g_opt_plan_rest_first = 0;   // dispatch the remainder pieces before the full ones (tuning)
static Option g_opt_plan_slots = 2048;     // wave slots the planner fills: 256 CUs x 4 SIMDs x 2 waves (tuning)
static Option g_opt_plan_xcd_affine = 1;   // batches: deal the work items so that an XCD sweeps one sightline at a time
static Option g_opt_plan_taper = 1;        // batches: pieces per block for the sightlines of the last round of eight (1 = off)
static Option g_opt_plan_affine_min_chunks = 256;  // ... the XCD-affine dealing only for sightlines at least this long (plan_host.h)
// launches of up to this many blocks of 16 samples finalise with 16 lanes per sample (kernels_common.h); 0: never
static Option g_opt_finalize_row16_blocks = DLAGP_FIN_ROW16_MAX_BLOCKS;
// read-only diagnostics of the most recent launch / sweep (dlagp_get_option): which kernel served it
static Option g_last_launch_path = 0;    // 1 generic kernel, 2 MFMA kernel
static Option g_last_launch_items = 0;   // work items (waves) of the last MFMA launch
static Option g_last_launch_parts = 0;   // most pieces a block of 16 samples was cut into
static Option g_last_sweep_cached = 0;   // the last multi-level sweep used the profile cache
static Option g_last_launch_patch = 0;   // core handling of the last MFMA launch (patch_mode: 0 in loop, 1, 2)
static Option g_last_launch_groups = 1;  // homogeneous launch groups the last call ran as (> 1: a batch of mixed flavours)

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
        const double a[DLAGP_WING_DEGREE + 1] = DLAGP_WING_COEF_INIT;
        const double y2 = lc.y[l] * lc.y[l];
        const double corr[6] = {0.0, 1.0, 5.0, 26.25, 0.0, 0.0};  // y^2 u (1 + 5u + 26.25u^2)
        for (int m = 0; m < 6; ++m)
            lc.wing[l][m] = m <= DLAGP_WING_DEGREE ? lc.k_wing[l] * (a[m] - y2 * corr[m]) : 0.0;
    }
    {
        static_assert(DLAGP_EXP_DEGREE == 9, "LineConstants::exp2c holds ten coefficients");
        const double e2[DLAGP_EXP_DEGREE + 1] = DLAGP_EXP2_COEF_INIT;
        for (int m = 0; m <= DLAGP_EXP_DEGREE; ++m) lc.exp2c[m] = e2[m];
    }
    lc.inv_sqrt2_sigma = 1.0 / sqrt2_sigma;
    lc.c = DLAGP_SPEED_OF_LIGHT_CGS;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(DLAGP_LINES), &lc, sizeof(lc)));
    // finalize_rank20 stages its partial sums in dynamic LDS
    for (const void *fn : {reinterpret_cast<const void *>(finalize_rank20<true, true>),
                           reinterpret_cast<const void *>(finalize_rank20<true, false>),
                           reinterpret_cast<const void *>(finalize_rank20<false, true>),
                           reinterpret_cast<const void *>(finalize_rank20<false, false>)})
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(DLAGP_FINALIZE_LDS_DOUBLES * sizeof(double))));
    g_device_ready[device] = 1;
    return DLAGP_OK;
}

// ------------------------------------------------------------------------------------------------
// device buffers
// ------------------------------------------------------------------------------------------------
// Released buffers are parked per (device, size) and handed out again: a catalogue run creates and
// destroys thousands of contexts of a few distinct sizes, and hipMalloc / hipFree (which synchronises the
// device) were a quarter of its wall time.  At most g_opt_pool_mb are parked (option "pool_mb", 0 = off);
// only buffers up to 64 MB take part (the big sweep workspaces stay with their launch group anyway).
static std::mutex g_pool_mutex;
static std::map<std::pair<int, size_t>, std::vector<void *>> g_pool;
static size_t g_pool_bytes = 0;
static Option g_opt_pool_mb = 2048;

static void pool_flush() {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    int current = 0;
    (void)hipGetDevice(&current);
    for (auto &kv : g_pool) {
        (void)hipSetDevice(kv.first.first);
        for (void *p : kv.second) (void)hipFree(p);
    }
    g_pool.clear();
    g_pool_bytes = 0;
    (void)hipSetDevice(current);
}

// buffers above the pool's size limit (sweep workspaces: partial sums, profile cache) that were allocated / freed since the
// library was loaded; a free synchronises the device (options "big_allocs", "big_frees": read-only diagnostics)
static Option g_stat_big_allocs = 0;
static Option g_stat_big_frees = 0;

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    static constexpr size_t kPoolMaxBuffer = (size_t)64 << 20;
    // Sizes are rounded up to eight steps per octave (<= 12.5 % more): the arrays of a spectrum scale with its pixel
    // count, which differs from spectrum to spectrum, and the pool hands out exact sizes only -- without the
    // rounding nearly every buffer of a new context was a hipMalloc and every released one a hipFree.
    static size_t bucket(size_t want) {
        if (want <= 4096) return 4096;
        size_t step = 512;
        while ((step << 4) < want) step <<= 1;  // step = 2^(floor(log2(want - 1)) - 3)
        return (want + step - 1) / step * step;
    }
    int reserve(size_t want) {
        if (want <= bytes) return DLAGP_OK;
        release();
        want = bucket(want);
        int device = 0;
        HIP_TRY(hipGetDevice(&device));
        if (want <= kPoolMaxBuffer) {
            std::lock_guard<std::mutex> lock(g_pool_mutex);
            auto it = g_pool.find({device, want});
            if (it != g_pool.end() && !it->second.empty()) {
                ptr = it->second.back();
                it->second.pop_back();
                g_pool_bytes -= want;
                bytes = want;
                return DLAGP_OK;
            }
        }
        HIP_TRY(hipMalloc(&ptr, want));
        if (want > kPoolMaxBuffer) g_stat_big_allocs.v.fetch_add(1, std::memory_order_relaxed);
        bytes = want;
        return DLAGP_OK;
    }
    void release() {
        if (!ptr) return;
        // a buffer that goes back to the pool (or to the allocator) must be idle: the caller has drained whatever stream
        // its last launch ran on.  Checked by the host-only sanitizer build's fake device (fake_device.h); nothing here
        // in the library proper.
        DLAGP_DEVICE_CHECK_IDLE(ptr);
        bool parked = false;
        if (bytes <= kPoolMaxBuffer) {
            int device = 0;
            if (hipGetDevice(&device) == hipSuccess) {
                std::lock_guard<std::mutex> lock(g_pool_mutex);
                if (g_pool_bytes + bytes <= (size_t)std::max<int64_t>(g_opt_pool_mb, 0) << 20) {
                    g_pool[{device, bytes}].push_back(ptr);
                    g_pool_bytes += bytes;
                    parked = true;
                }
            }
        }
        if (!parked) {
            (void)hipFree(ptr);  // synchronises the device
            if (bytes > kPoolMaxBuffer) g_stat_big_frees.v.fetch_add(1, std::memory_order_relaxed);
        }
        ptr = nullptr;
        bytes = 0;
    }
    template <typename T>
    T *as() const {
        return static_cast<T *>(ptr);
    }
};

// Page-locked host memory for the host-pointer entry points.  hipMemcpy to or from ordinary (pageable)
// caller memory makes the runtime pin and later unpin the caller's pages; for freshly allocated arrays
// of a few MB (NumPy results) that left ~20 ms of runtime work behind every sweep, paid by the next call.
// Transfers of more than 64 KB therefore go through this buffer (one extra CPU copy, ~10 GB/s).
struct PinnedBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t want) {
        if (want <= bytes) return DLAGP_OK;
        release();
        HIP_TRY(hipHostMalloc(&ptr, want, hipHostMallocDefault));
        bytes = want;
        return DLAGP_OK;
    }
    void release() {
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
};

// scratch shared by the launches of one launch group
struct Workspace {
    DeviceBuffer part;    // partial rows [nsplit][rows][row_width]
    DeviceBuffer part2;   // ranks above 20 on the MFMA path: the column groups' sums regrouped into one row per block
    DeviceBuffer status;  // int flag raised by finalize on a non-PD pivot
    DeviceBuffer null_part, null_status;  // the same two for the null model's side launch (LaunchGroup::side_stream)
    DeviceBuffer z, nhi, out;  // staging for the host-pointer entry points
    // device-resident occupation sweep: raw level log-likelihoods, cdf scratch, per-sightline alive flag,
    // and staging of its host-pointer variant
    DeviceBuffer level_ll, cdf, alive, uniforms, evidences, sample_ll, base;
    DeviceBuffer level_x;  // what the phases of the level finish hand each other, [nb][DLAGP_LEVEL_EXCH] (kernels_common.h)
    DeviceBuffer ev_part, ev_tickets;  // the evidence tail of finalize_rank20: the blocks' triples, the sightlines' ticket counters
    DeviceBuffer ties;     // [nb][max_dlas - 1] searches of the resampling that ended within DLAGP_TIE_MARGIN of a cdf step
    DeviceBuffer cache;  // profile cache of the multi-level sweep, [nb][S][max n_slots]
    DeviceBuffer sub_z, sub_nhi, sub_evidences, sub_ll;  // the one-absorber model of dlagp_batch_model_evidences
    DeviceBuffer sub_level_ll, sub_alive;  // pair launches: the one-absorber model's raw level and its per-sightline flags
    DeviceBuffer sub_cdf, sub_level_x;     // ... and its own weights and exchange: its level finish rides beside the sweep's level 1
    PinnedBuffer stage;           // host staging of the host-pointer entry points
    DeviceBuffer pack;            // ... and its device-side twin: small transfers cross the bus as one copy (Transfers)
    void release() {
        stage.release();
        pack.release();
        cache.release();
        level_ll.release();
        cdf.release();
        level_x.release();
        ev_part.release();
        ev_tickets.release();
        ties.release();
        alive.release();
        uniforms.release();
        evidences.release();
        sample_ll.release();
        base.release();
        part.release();
        part2.release();
        status.release();
        null_part.release();
        null_status.release();
        z.release();
        nhi.release();
        out.release();
        sub_z.release();
        sub_nhi.release();
        sub_evidences.release();
        sub_ll.release();
        sub_level_ll.release();
        sub_alive.release();
        sub_cdf.release();
        sub_level_x.release();
    }
};

static std::mutex g_spare_mutex;
// per device: workspaces of destroyed batches, adopted by the next batches created.  A pipelined catalogue run has
// up to five batches alive at a time (one being prepared, three in their sweeps, one being finished): with a single
// spare every other batch freed and reallocated gigabytes (hipFree synchronises the device: 7 ms per batch).
static std::map<int, std::vector<Workspace>> g_spare_workspace;
static Option g_opt_spare_workspaces = 5;
static Option g_opt_shared_work_stream = 1;
static Option g_opt_shared_rows = 1;  // cached levels: a wave whose samples stack the same cached row reads it once
// test hook: contexts with a negative noise variance stay on the MFMA kernel (default: they take the generic kernel, whose
// per-pixel log(d) is NaN for any negative d as in the reference) -- the tests of that kernel's non-PD flag need them
static Option g_opt_negative_noise_mfma = 0;
static Option g_opt_fold_evidence = 1;  // dlagp_*eval_evidence_device: the evidence in finalize_rank20's tail (0: a launch of its own)
static Option g_opt_finalize_wave = 1;  // ranks above 20: one wave per sample factors the matrix in LDS (0: one thread per sample)

// Decomposition of one MFMA launch (a launch group at one sample count) into work items, see build_plan
struct LaunchPlan {
    // one device allocation, one upload: [PassDesc passes[] | WorkItem items[] (largest first) | unsigned char counts[]]
    //   passes  the pass lists of all items (shared between items with equal pixel ranges)
    //   counts  pieces per block of 16 samples (partial-sum slots to add), [nblocks]
    DeviceBuffer blob;
    size_t items_at = 0, counts_at = 0;
    int nitems = 0;
    int max_parts = 1;
    const PassDesc *passes() const { return static_cast<const PassDesc *>(blob.ptr); }
    const WorkItem *items() const { return reinterpret_cast<const WorkItem *>(static_cast<const char *>(blob.ptr) + items_at); }
    const unsigned char *counts() const { return static_cast<const unsigned char *>(blob.ptr) + counts_at; }
    void release() { blob.release(); }
};

struct dlagp_ctx;

// one or more sightlines evaluated by one launch sequence
struct LaunchGroup {
    int device = 0;
    std::vector<dlagp_ctx *> ctxs;
    DeviceBuffer d_table;  // SightlineDev[nb]
    // a batch of ONE member reads the member's own table (its context's group holds the same descriptor at position 0): no
    // second upload, no wait for it.  Not owned; a batch whose member has been destroyed no longer launches (batch_alive).
    const SightlineDev *table_alias = nullptr;
    const SightlineDev *table() const { return table_alias ? table_alias : static_cast<const SightlineDev *>(d_table.ptr); }
    int k = 0, row_width = 0, num_lines = 0;
    bool broadening = true;
    int max_entries = 0, max_npad = 0;
    hipStream_t last_stream = nullptr;  // stream of the most recent launch (drained before buffers are recycled)
    bool launched = false;
    // the null model of a model-evidence call runs BESIDE the sweeps, on the calling thread's transfer stream, with partial
    // sums and a status flag of its own (ws.null_part, ws.null_status): it depends on nothing the sweeps produce
    hipStream_t side_stream = nullptr;
    bool side_launched = false;
    // a launch plan's work items that went up with a call's uploads (dlagp_batch_model_evidences_start): the copy runs on the
    // calling thread's transfer stream; the call's kernels wait for it, but a call that FAILS before its first launch leaves
    // it in flight with nothing of the group behind it -- and plans are released with the group (found by the sanitizer replay)
    hipStream_t upload_stream = nullptr;
    bool upload_pending = false;
    bool mfma_ok = false;
    int col_groups = 1;         // MFMA path: column groups per block (1 for rank <= 20, see kernels_mfma.h)
    bool grid_uniform = false;  // every member's wavelength grid is log-uniform (core patch eligible)
    bool lls_break = false;     // members carry the Lyman-limit term (all or none, else no MFMA path)
    Workspace ws;
    int max_slots = 0;  // largest n_slots of a member (row stride of the profile cache)
    // key: (S, and the planner options in force: forced split count, item cost, dispatch order, slot count)
    typedef std::tuple<int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t> PlanKey;
    std::map<PlanKey, std::unique_ptr<LaunchPlan>> plans;
    // A batch whose members differ in line count, broadening or Lyman-limit flag cannot share one instantiation of
    // the MFMA kernel.  Instead of sending all of them to the generic kernel (21x slower) it runs as several
    // homogeneous launch groups over subsets of its members, every kernel reading samples and writing results at
    // the member's position in the caller's arrays (SightlineDev::batch_index).  Empty for a homogeneous batch.
    // This group then keeps the scratch that spans all members (staging, per-level likelihoods, the shared non-PD
    // flag); the parts keep their own partial sums, launch plans and profile caches.
    std::vector<std::unique_ptr<LaunchGroup>> parts;
    // kernels in flight on the group's (and its parts') most recent stream are done on return
    void drain() {
        if (launched) (void)hipStreamSynchronize(last_stream);
        launched = false;
        if (side_launched) (void)hipStreamSynchronize(side_stream);
        side_launched = false;
        if (upload_pending) (void)hipStreamSynchronize(upload_stream);
        upload_pending = false;
        for (auto &p : parts) p->drain();
    }
    void release() {
        // asynchronous entry points may have left kernels in flight that still use these buffers, and released
        // buffers go back to a pool other contexts draw from
        drain();
        for (auto &p : parts) p->release();
        parts.clear();
        d_table.release();
        table_alias = nullptr;
        ws.release();
        for (auto &kv : plans) kv.second->release();
        plans.clear();
    }
};

// Every growth of a launch group's scratch goes through here.  The asynchronous entry points leave kernels in flight
// on the caller's stream, and a buffer that is replaced goes back to a pool other contexts (other host threads) draw
// from: the stream of the group's most recent launch is drained before a live buffer is let go (the stream rule of
// include/dlagp.h: "a call that needs larger scratch drains that stream first").
static int grow(LaunchGroup &g, DeviceBuffer &buf, size_t want) {
    if (want <= buf.bytes) return DLAGP_OK;
    if (buf.ptr && g.launched) {
        HIP_TRY(hipStreamSynchronize(g.last_stream));
        g.launched = false;  // nothing of this group is in flight any more
    }
    return buf.reserve(want);
}

// A launch group's scratch is shared by its launches, which are ordered only within one stream.  A call that arrives on
// another stream than the group's previous launches (e.g. a host-pointer call, which runs on the calling thread's own
// stream, after `*_device` calls on the caller's stream) first drains the previous stream.
static int enter_stream(LaunchGroup &g, hipStream_t stream) {
    if (g.launched && g.last_stream != stream) {
        HIP_TRY(hipStreamSynchronize(g.last_stream));
        g.launched = false;
    }
    for (auto &p : g.parts)
        if (p->launched && p->last_stream != stream) {
            HIP_TRY(hipStreamSynchronize(p->last_stream));
            p->launched = false;
        }
    return DLAGP_OK;
}

// learned GP model resident on one device (state of NullGP.__init__, reference null_gp.py:35-71)
struct dlagp_model {
    int device = 0;
    ModelDev dev{};
    DeviceBuffer grid, mu, log_omega, M;
    double grid_min = 0.0, grid_max = 0.0;
};

struct dlagp_ctx {
    int device = 0;
    int user_k = 0;            // rank the caller passed (k <= 20 is zero-padded to 20 internally)
    SightlineDev sl{};         // host copy of the descriptor (device pointers inside)
    DeviceBuffer y, mu, omega2, v, pos, lam, P, aux;
    DeviceBuffer Pm, slot_pixel, slot_pos;  // MFMA path: slot-ordered operand and slot tables
    DeviceBuffer dense_M;      // interpolated / uploaded factor, row-major [n][k], kept for dlagp_ctx_read_model
    DeviceBuffer scratch_a, scratch_b;  // inputs of the set-up kernels, released once the set-up stream has drained
    bool setup_pending = false;         // created with DLAGP_FLAG_DEFER_SYNC and not known to be complete yet
    std::mutex setup_mutex;             // two threads may first-use one deferred context at once (two batches sharing a member)
    hipEvent_t setup_event = nullptr;   // ... recorded behind the set-up work on the creating thread's set-up stream
    bool negative_noise = false;        // v or omega2 has a negative entry: d = omega2 a^2 + v may be negative
    std::vector<int> h_pos;    // host copy of pos[]
    std::vector<int> h_slot_pos;  // window start of every slot (padding slots repeat their chunk's last pixel), for planning
    bool mfma_ok = false;      // the MFMA kernel can serve this sightline
    int col_groups = 1;        // ... with this many column groups (1: rank <= 20, padded to 20)
    LaunchGroup self;          // the group {this}
    std::vector<dlagp_batch *> batches;  // live batches this context is a member of (g_membership_mutex)
};

struct dlagp_batch {
    LaunchGroup group;
    // near-tie counts of the most recent sweep (dlagp_batch_resample_ties): with the results of a host-pointer sweep they are
    // already here; after a *_device sweep they still sit in the workspace, behind the sweep's kernels on `ties_stream`
    std::vector<int32_t> ties;
    int ties_levels = -1;  // max_dlas - 1 of that sweep; -1: no sweep yet
    bool ties_on_device = false;
    hipStream_t ties_stream = nullptr;
    // dlagp_batch_model_evidences_start has run and _finish has not
    struct Pending {
        bool active = false;
        int64_t S = 0, nhi_stride = 0;
        int max_dlas = 0;
        double min_z_separation = 0.0;
        size_t stage_used = 0;          // staging bytes of the first phase: the second phase's lie behind them
        hipStream_t stream = nullptr;   // the calling thread's kernel stream: _finish must come from the same thread
        bool null_on_side = false;      // the null model runs beside the sweeps (launch_null_side) ...
        bool null_launched = false;     // ... and has been enqueued (by _start, or by _finish beside the deepest level's finish)
        bool pair = false;              // the sub-DLA sweep rode in level 1's launch (pair launch): _finish resumes that sweep
        int64_t sub_nhi_stride = 0;
        int cached = 0;                 // ... and whether it keeps level-1 profiles (decided in _start)
    } pending;
};

// which batches a context belongs to: a context destroyed before its batches drains their streams first
static std::mutex g_membership_mutex;

// ------------------------------------------------------------------------------------------------
// planning (MFMA path): plan_host.h -- pure host code, also built alone under ASan/UBSan and fuzzed on the CPU
// ------------------------------------------------------------------------------------------------
static plan::Options planner_options(int column_groups = 1) {
    plan::Options opt;
    opt.column_groups = column_groups;
    opt.pixel_splits = g_opt_pixel_splits;
    opt.item_cost = g_opt_plan_item_cost;
    opt.rest_first = g_opt_plan_rest_first;
    opt.slots = g_opt_plan_slots;
    opt.xcd_affine = g_opt_plan_xcd_affine;
    opt.taper = g_opt_plan_taper;
    opt.affine_min_chunks = g_opt_plan_affine_min_chunks;
    return opt;
}

// `defer_upload`: a NEW plan's bytes are handed to the caller instead of being copied here -- it sends them to (*out)->blob.ptr
// with the call's other uploads, in front of the launch (dlagp_batch_model_evidences_start); left empty when the plan existed.
static int build_plan(LaunchGroup &g, int64_t S, LaunchPlan **out, std::vector<char> *defer_upload = nullptr) {
    const LaunchGroup::PlanKey key(S, g_opt_pixel_splits, g_opt_plan_item_cost, g_opt_plan_rest_first, g_opt_plan_slots,
                                   g_opt_plan_xcd_affine, g_opt_plan_taper, g_opt_plan_affine_min_chunks);
    auto it = g.plans.find(key);
    if (it != g.plans.end()) {
        *out = it->second.get();
        return DLAGP_OK;
    }
    std::vector<plan::Sightline> members;
    members.reserve(g.ctxs.size());
    for (const dlagp_ctx *c : g.ctxs)
        members.push_back(plan::Sightline{&c->h_slot_pos, c->sl.ntaps, (long)(c->sl.n_slots / DLAGP_CHUNK_PIXELS)});
    plan::HostPlan host;
    std::string err;
    if (plan::build(members, S, planner_options(g.col_groups), host, err)) return fail(DLAGP_EINVAL, "%s", err.c_str());
    // One plan per distinct (S, planner options): a by-sample run adds one per slice size, a tuning session one per
    // option set.  Beyond kMaxPlans the cache starts over -- after draining the group's stream, since a launch in
    // flight still reads its plan.
    constexpr size_t kMaxPlans = 16;
    if (g.plans.size() >= kMaxPlans) {
        if (g.launched) {
            HIP_TRY(hipStreamSynchronize(g.last_stream));
            g.launched = false;
        }
        for (auto &kv : g.plans) kv.second->release();
        g.plans.clear();
    }
    std::unique_ptr<LaunchPlan> plan(new LaunchPlan());
    // planning runs on the launching thread, ahead of the launch on the caller's stream: ONE blocking copy (three cost
    // a one-spectrum call 50 us more) -- or none, when the caller has uploads of its own under way (defer_upload)
    const size_t passes_bytes = host.passes.size() * sizeof(PassDesc), items_bytes = host.items.size() * sizeof(WorkItem);
    plan->items_at = (passes_bytes + 255) & ~(size_t)255;
    plan->counts_at = (plan->items_at + items_bytes + 255) & ~(size_t)255;
    std::vector<char> staged((plan->counts_at + std::max<size_t>(1, host.counts.size()) + 15) & ~(size_t)15, 0);
    std::memcpy(staged.data(), host.passes.data(), passes_bytes);
    std::memcpy(staged.data() + plan->items_at, host.items.data(), items_bytes);
    std::memcpy(staged.data() + plan->counts_at, host.counts.data(), host.counts.size());
    int rc = plan->blob.reserve(staged.size());
    if (rc) return rc;
    if (defer_upload) defer_upload->swap(staged);
    else HIP_TRY(hipMemcpy(plan->blob.ptr, staged.data(), staged.size(), hipMemcpyHostToDevice));
    plan->nitems = (int)host.items.size();
    plan->max_parts = host.max_parts;
    *out = plan.get();
    g.plans[key] = std::move(plan);
    return DLAGP_OK;
}

// ------------------------------------------------------------------------------------------------
// launch sequence common to single-sightline and batch evaluation
// ------------------------------------------------------------------------------------------------
// The timing events (option "timing") ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL): recorded with
// hipEventRecord before and after the launch they were two barrier packets of ~6 us each around a 0.29 ms kernel.
#define DLAGP_LAUNCH_MFMA(...)                                                                                     \
    hipExtLaunchKernelGGL((accumulate_mfma<__VA_ARGS__>), grid, dim3(DLAGP_MFMA_THREADS), (uint32_t)prep_lds, stream, \
                          ev_start, ev_stop, 0u, table,                                                             \
                       plan->passes(), plan->items(), nb, S, kd, src, part, \
                       status, launch_flags, fused_out, fused_out2)

// how the line cores are handled (kernels_mfma.h): 0 tested for inside the sample loop; 1 one absorber per sample,
// repaired once per wave; 2 several absorbers per sample, repaired pass by pass.  1 and 2 need three lines and a
// log-uniform wavelength grid.
static int patch_mode(const LaunchGroup &g, int kd) {
    if (g.num_lines != 3 || !g.grid_uniform || call_options().core_patch == 0 || kd < 1) return 0;
    return kd == 1 ? 1 : 2;
}

template <int LINES, bool BROADEN>
static void launch_mfma(dim3 grid, hipStream_t stream, const SightlineDev *table, const LaunchPlan *plan, int nb,
                        long S, int kd, int num_lines, const SampleSource &src, double *part, int *status,
                        int patch, bool lls, int launch_flags, hipEvent_t ev_start, hipEvent_t ev_stop, double *fused_out,
                        double *fused_out2) {
    // 16 samples + one spare row that the look-ahead of the sample loop may read; PATCH 2: 3 ints per (sample, absorber)
    const size_t prep_lds = (size_t)(DLAGP_MFMA_GROUP + 1) * kd * (lls ? 6 : (LINES > 0 ? LINES : num_lines) + 1) * sizeof(double) +
                            (patch == 2 ? (size_t)DLAGP_MFMA_GROUP * kd * 3 * sizeof(int) : 0);
    if constexpr (LINES == 3) {
        if (lls) {  // Lyman-limit systems (voigt_lls.py): the continuum term rides in the 3-line sample loop
            if (patch == 1 && src.cache_mode == 1) DLAGP_LAUNCH_MFMA(3, BROADEN, false, 1, 1, true);
            else if (patch == 1) DLAGP_LAUNCH_MFMA(3, BROADEN, false, 0, 1, true);
            else if (patch == 2) DLAGP_LAUNCH_MFMA(3, BROADEN, false, 0, 2, true);
            else if (src.cache_mode == 1) DLAGP_LAUNCH_MFMA(3, BROADEN, false, 1, 0, true);
            else DLAGP_LAUNCH_MFMA(3, BROADEN, false, 0, 0, true);
            return;
        }
        // the production flavours (3 lines, broadening) also exist with the wave's own finalisation (fused_eligible)
        if (patch == 1) {  // line cores repaired outside the sample loop (kd == 1, uniform grid)
            if constexpr (BROADEN) {
                if (src.pair_S) {  // pair launch (two one-absorber models, shared optical depth)
                    if (fused_out && src.cache_mode == 1) DLAGP_LAUNCH_MFMA(3, true, false, 1, 1, false, true, true);
                    else if (fused_out) DLAGP_LAUNCH_MFMA(3, true, false, 0, 1, false, true, true);
                    else if (src.cache_mode == 1) DLAGP_LAUNCH_MFMA(3, true, false, 1, 1, false, false, true);
                    else DLAGP_LAUNCH_MFMA(3, true, false, 0, 1, false, false, true);
                    return;
                }
                if (fused_out && src.cache_mode == 1) {
                    DLAGP_LAUNCH_MFMA(3, true, false, 1, 1, false, true);
                    return;
                }
                if (fused_out) {
                    DLAGP_LAUNCH_MFMA(3, true, false, 0, 1, false, true);
                    return;
                }
            }
            if (src.cache_mode == 1) DLAGP_LAUNCH_MFMA(3, BROADEN, false, 1, 1);
            else if (call_options().debug & 2) DLAGP_LAUNCH_MFMA(3, BROADEN, true, 0, 1);  // with per-phase cycle stamps
            else DLAGP_LAUNCH_MFMA(3, BROADEN, false, 0, 1);
            return;
        }
        if (patch == 2 && !(call_options().debug & 2)) {  // several absorbers per sample, cores repaired pass by pass
            if constexpr (BROADEN) {
                if (fused_out) {
                    DLAGP_LAUNCH_MFMA(3, true, false, 0, 2, false, true);
                    return;
                }
            }
            DLAGP_LAUNCH_MFMA(3, BROADEN, false, 0, 2);
            return;
        }
    }
    if (src.cache_mode == 1) DLAGP_LAUNCH_MFMA(LINES, BROADEN, false, 1);  // level 1 of a cached sweep: also stores the profiles
    else if (call_options().debug & 2) DLAGP_LAUNCH_MFMA(LINES, BROADEN, true);     // diagnostic instantiation with per-phase cycle stamps
    else DLAGP_LAUNCH_MFMA(LINES, BROADEN, false);
}

// whether a launch with kd absorbers per sample runs the MFMA kernel (else the generic one); `cached`:
// the launch multiplies cached profiles (levels > 1 of a sweep) and evaluates no Voigt profile itself
static bool mfma_path_serves(const LaunchGroup &g, int kd, bool cached = false) {
    // the MFMA kernel keeps its 16 samples' per-absorber constants in dynamic LDS
    const size_t mfma_prep_lds = (size_t)(DLAGP_MFMA_GROUP + 1) * std::max(kd, 1) * ((g.lls_break ? 5 : g.num_lines) + 1 + 2) * sizeof(double);
    if (g.lls_break && !cached && g.num_lines != 3) return false;  // the Lyman-limit term lives in the 3-line sample loop
    return g.mfma_ok && kd >= 1 && mfma_prep_lds <= 40 * 1024 && call_options().path != 1;
}

// Whether the accumulate waves of a launch finish their own blocks (kernels_mfma.h, FUSED): every block is ONE piece (the
// plans of batches: more blocks than wave slots), rank 20, and the flavour is one of the production instantiations
// that exist in that form -- 3 lines with broadening, no Lyman-limit term, the line cores repaired outside the sample
// loop; the cached levels of a sweep.  Option "fuse_finalize" = 0 keeps the two-kernel form (the cross-check).
static Option g_last_launch_fused = 0;
static bool fused_eligible(const LaunchGroup &g, const LaunchPlan *plan, int patch, int cache_mode) {
    if (!call_options().fuse_finalize || !plan || plan->max_parts != 1 || g.col_groups != 1 || g.k != DLAGP_FIN_K) return false;
    if (call_options().debug & 2) return false;  // the stamped diagnostic instantiation is unfused
    if (cache_mode == 2) return true;
    return g.num_lines == 3 && g.broadening && !g.lls_break && (patch == 1 || (patch == 2 && cache_mode == 0));
}

// Pair launches (SampleSource::pair_S): the sub-DLA sweep and level 1 of the DLA sweep in ONE launch of 16 ceil(S / 8)
// virtual samples when both models sample the same redshifts.  Needs the one-absorber production kernel (3 lines,
// broadening, cores repaired once per wave, rank 20, a homogeneous batch); finished by the waves themselves on single-piece
// plans and by finalize_rank20 otherwise.  Option "pair_models" = 0 keeps the two launches (the cross-check).
static Option g_last_sweep_paired = 0;
static Option g_last_evidence_folded = 0;  // the most recent dlagp_*eval_evidence_device: 1 = evidence from finalize_rank20's tail
static int64_t pair_virtual_samples(int64_t S) { return 16 * ((S + 7) / 8); }
static bool pair_eligible(LaunchGroup &g, int64_t S) {
    if (!call_options().pair_models || !g.parts.empty() || (call_options().debug & 2)) return false;
    return mfma_path_serves(g, 1) && patch_mode(g, 1) == 1 && g.broadening && !g.lls_break && g.col_groups == 1 &&
           g.k == DLAGP_FIN_K && S >= 1;
}

// finalisation of a general rank (above 20) from column-major rows: one wave per sample, or (option finalize_wave = 0,
// the cross-check) one thread per sample
static void launch_finalize_general(hipStream_t stream, const double *part, int nsplit, long rows, long S,
                                    const SightlineDev *table, double *d_out, int *status, const unsigned char *counts) {
    if (g_opt_finalize_wave) {
        const size_t lds = (size_t)DLAGP_FINW_SAMPLES * finalize_wave_lds_doubles(DLAGP_MAX_RANK) * sizeof(double);
        hipLaunchKernelGGL(finalize_wave, dim3((unsigned)((rows + DLAGP_FINW_SAMPLES - 1) / DLAGP_FINW_SAMPLES)),
                           dim3(64 * DLAGP_FINW_SAMPLES), lds, stream, part, nsplit, rows, S, table, d_out, status, counts);
    } else {
        const int threads = DLAGP_FINALIZE_THREADS;
        hipLaunchKernelGGL(finalize_samples<0>, dim3((unsigned)((rows + threads - 1) / threads)), dim3(threads), 0, stream,
                           part, nsplit, rows, S, table, d_out, status, counts);
    }
}

// One homogeneous launch group: accumulate + finalize.  `status`: the non-PD flag the finalisation raises (the
// group's own unless the caller shares one between the parts of a heterogeneous batch); `keep_status`: this launch
// must not clear it (an earlier part of the same level may have raised it).
// `tail`: the level evidence the caller wants with the log-likelihoods (dlagp_eval_evidence_device).  A plan that ends in
// finalize_rank20 computes it there (EvidenceTail, dlagp_device.h) and sets tail->done; any other plan leaves it to the caller.
struct TailRequest {
    int level = 0;
    const unsigned char *too_close = nullptr;
    double *evidence = nullptr;
    bool done = false;
};
static int launch_one(LaunchGroup &g, int64_t S, int kd, const SampleSource &src, double *d_out, hipStream_t stream,
                      int *status_shared = nullptr, bool keep_status = false, double *d_out2 = nullptr,
                      TailRequest *tail = nullptr) {
    CallOptionsScope options_of_this_call;
    if (S <= 0) return DLAGP_OK;
    Workspace &ws = g.ws;
    const int nb = (int)g.ctxs.size();
    const long rows = (long)nb * S;
    const SightlineDev *table = g.table();
    int rc;
    if ((rc = enter_stream(g, stream))) return rc;

    bool use_mfma = mfma_path_serves(g, kd, src.cache_mode == 2);
    if (call_options().path == 2 && kd >= 1 && !use_mfma)
        return fail(DLAGP_EINVAL, "MFMA path requested but this group is not eligible (negative noise variance, mixed line counts / broadening / Lyman-limit flags, or kd x lines too large)");
    if (!use_mfma && src.cache_mode)
        return fail(DLAGP_EINVAL, "internal: the profile cache is only served by the MFMA path");
    LaunchPlan *plan = nullptr;
    if (use_mfma && (rc = build_plan(g, S, &plan))) return rc;
    // the null model on the generic kernel (one sample per sightline): four pixel ranges per sightline, always four,
    // so that a sightline's sums are formed the same way alone and in any batch
    const int nparts = plan ? plan->max_parts : (kd == 0 ? 4 : 1);

    // partial sums: [piece][sightline x ceil(S/16) blocks][row_width columns][16 samples]
    const size_t nblocks = (size_t)nb * ((S + 15) / 16);
    // ranks above 20 on the MFMA path: every column group leaves rank-20-sized sums, regrouped into ws.part2 (below)
    const bool grouped = use_mfma && g.col_groups > 1;
    const int patch = !use_mfma || src.cache_mode == 2 ? 0 : patch_mode(g, kd);
    // single-piece plans: the accumulate waves finish their own blocks, no partial sums cross HBM (and none are allocated)
    const bool fused = use_mfma && fused_eligible(g, plan, patch, src.cache_mode);
    if (src.pair_S && !(use_mfma && patch == 1 && d_out2 && kd == 1 && g.num_lines == 3 && g.broadening && !g.lls_break &&
                        g.col_groups == 1 && g.k == DLAGP_FIN_K && !(call_options().debug & 2)))
        return fail(DLAGP_EINVAL, "internal: a pair launch needs the one-absorber 3-line MFMA kernel (see pair_eligible)");
    const size_t part_bytes = fused ? 0
                              : grouped ? (size_t)g.col_groups * nparts * nblocks * DLAGP_MFMA_ROW_WIDTH * 16 * sizeof(double)
                                        : (size_t)nparts * nblocks * g.row_width * 16 * sizeof(double);
    if (grouped && (rc = grow(g, ws.part2, nblocks * g.row_width * 16 * sizeof(double)))) return rc;
    // kernels enqueued earlier may still be using the buffer that a larger request replaces: grow() drains them
    if ((rc = grow(g, ws.part, part_bytes))) return rc;
    if ((rc = grow(g, ws.status, sizeof(int)))) return rc;
    g.last_stream = stream;
    g.launched = true;

    // timing (option "timing"): a pair of events attached to the accumulate launch.  The pair joins the list that
    // dlagp_timing_read reads only once the launch has been enqueued; any earlier exit destroys it.
    struct TimingGuard {
        EventPair ev{};
        bool armed = false;
        ~TimingGuard() {
            if (!armed) return;
            (void)hipEventDestroy(ev.start);
            (void)hipEventDestroy(ev.stop);
        }
    } tg;
    const bool timing = g_opt_timing != 0;
    if (timing) {
        HIP_TRY(hipEventCreate(&tg.ev.start));
        if (hipEventCreate(&tg.ev.stop) != hipSuccess) {
            (void)hipEventDestroy(tg.ev.start);
            return fail(DLAGP_EDEVICE, "hipEventCreate failed");
        }
        tg.armed = true;
    }
    const hipEvent_t ev_start = timing ? tg.ev.start : nullptr, ev_stop = timing ? tg.ev.stop : nullptr;
    // With several host threads launching onto ONE stream (shared_work_stream) the runtime enqueues the start marker and the
    // kernel as two commands: another thread's kernel that slips in between is counted into this launch's interval
    // (three threads: the sum of the intervals came out 14 % above the wall time's share, profiles/r05_experiments.txt
    // item 23).  While timing is on, launches therefore enqueue one at a time.
    // (one lock per device: launches on different devices never share a stream.  The timed catalogue figures of
    // profiles/r05_catalogue_pipeline*.json and r06_* were therefore taken with a batch's launches enqueued strictly one
    // after the other -- a slightly different enqueue order than production's, which takes no lock.)
    static std::mutex timing_order[16];
    std::unique_lock<std::mutex> one_at_a_time(timing_order[(unsigned)g.device % 16], std::defer_lock);
    if (timing) one_at_a_time.lock();

    int *status = status_shared ? status_shared : ws.status.as<int>();
    g_last_launch_path = use_mfma ? 2 : 1;
    g_last_launch_fused = fused ? 1 : 0;
    if (use_mfma) {
        g_last_launch_items = plan->nitems;
        g_last_launch_parts = plan->max_parts;
        dim3 grid((unsigned)plan->nitems);
        const bool lines3 = g.num_lines == 3;
        g_last_launch_patch = patch;
        double *fused_out = fused ? d_out : nullptr;
        double *fused_out2 = src.pair_S ? d_out2 : nullptr;
        // the waves raise the non-PD flag themselves: cleared here, in front of the launch (the unfused kernel's first
        // workgroup clears it, long before the finalisation kernel can raise it)
        if (fused && !keep_status) HIP_TRY(hipMemsetAsync(status, 0, sizeof(int), stream));
        const bool lls = g.lls_break;
        double *part = ws.part.as<double>();
        const int launch_flags = (keep_status ? DLAGP_LAUNCH_KEEP_STATUS : 0) | (g_opt_shared_rows ? 0 : DLAGP_LAUNCH_GENERAL_ROWS) |
                                 (nparts & DLAGP_LAUNCH_PIECES_MASK);
        if (src.cache_mode == 2) {
            // [kd - 1][16] stacked rows of the wave's samples + its pool of 16 distinct rows (ints), see accumulate_mfma_body
            const size_t prep_lds = (size_t)(DLAGP_MAX_ABSORBERS * 16) * sizeof(int);
            if (fused) DLAGP_LAUNCH_MFMA(3, true, false, 2, 0, false, true);
            else DLAGP_LAUNCH_MFMA(3, true, false, 2);
            (void)fused_out2;
        } else if (g.broadening) {
            if (lines3) launch_mfma<3, true>(grid, stream, table, plan, nb, S, kd, g.num_lines, src, part, status, patch, lls, launch_flags, ev_start, ev_stop, fused_out, fused_out2);
            else launch_mfma<0, true>(grid, stream, table, plan, nb, S, kd, g.num_lines, src, part, status, false, false, launch_flags, ev_start, ev_stop, fused_out, fused_out2);
        } else {
            if (lines3) launch_mfma<3, false>(grid, stream, table, plan, nb, S, kd, g.num_lines, src, part, status, patch, lls, launch_flags, ev_start, ev_stop, fused_out, fused_out2);
            else launch_mfma<0, false>(grid, stream, table, plan, nb, S, kd, g.num_lines, src, part, status, false, false, launch_flags, ev_start, ev_stop, fused_out, fused_out2);
        }
        HIP_TRY(hipGetLastError());
    } else {
        // w[] (n_pad values) reuses the raw-profile buffer: size it for both
        const int lds_entries = std::max(g.max_entries, g.max_npad);
        const int lds_pixels = g.max_npad;
        const size_t lds_bytes = (size_t)(lds_entries + lds_pixels + DLAGP_MAX_LINES + 1 + 8) * sizeof(double);
        if (lds_bytes > 160 * 1024)
            return fail(DLAGP_EINVAL, "sightline too long for the generic kernel: %zu B of LDS needed", lds_bytes);
        if (lds_bytes > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(accumulate_generic),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        dim3 grid((unsigned)S, (unsigned)nb, (unsigned)nparts);
        hipExtLaunchKernelGGL(accumulate_generic, grid, dim3(DLAGP_GENERIC_THREADS), (uint32_t)lds_bytes, stream,
                              ev_start, ev_stop, 0u, table, (long)S, kd, src, ws.part.as<double>(), status, lds_entries,
                              lds_pixels, keep_status ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }

    if (timing) {  // both events belong to the launch that has just been enqueued
        std::lock_guard<std::mutex> lock(g_timing_mutex);
        g_timing_events.push_back(tg.ev);  // owned by the list from here on: dlagp_timing_read destroys them
        tg.armed = false;
    }

    if (fused) {
        // nothing to finish: the accumulate waves wrote d_out
    } else if (grouped) {
        hipLaunchKernelGGL(regroup_native, dim3((unsigned)nblocks), dim3(256), 0, stream, ws.part.as<const double>(), nparts,
                           (long)S, table, plan->counts(), ws.part2.as<double>());
        launch_finalize_general(stream, ws.part2.as<const double>(), 1, rows, (long)S, table, d_out, status, nullptr);
    } else if (g.k == DLAGP_FIN_K) {
        const size_t lds_bytes = DLAGP_FINALIZE_LDS_DOUBLES * sizeof(double);
        const bool row16 = (int64_t)nblocks <= g_opt_finalize_row16_blocks;
        // which layout the partial sums are in: the MFMA kernel's tiles as they come, or the generic kernel's columns
        auto fin = row16 ? (use_mfma ? finalize_rank20<true, true> : finalize_rank20<true, false>)
                         : (use_mfma ? finalize_rank20<false, true> : finalize_rank20<false, false>);
        EvidenceTail ev{};
        if (tail && tail->evidence && !src.pair_S && g_opt_fold_evidence) {
            // the blocks' triples and the sightlines' ticket counters (zero between launches: zeroed when the buffer is
            // made, put back by the kernel's last workgroup)
            if ((rc = grow(g, ws.ev_part, (size_t)nblocks * 3 * sizeof(double)))) return rc;
            if (ws.ev_tickets.bytes < (size_t)nb * sizeof(int)) {
                if ((rc = grow(g, ws.ev_tickets, (size_t)nb * sizeof(int)))) return rc;
                HIP_TRY(hipMemsetAsync(ws.ev_tickets.ptr, 0, ws.ev_tickets.bytes, stream));
            }
            ev = EvidenceTail{tail->evidence, tail->too_close, ws.ev_part.as<double>(), ws.ev_tickets.as<int>(),
                              std::log((double)S), tail->level};
            tail->done = true;
        }
        hipLaunchKernelGGL(fin, dim3((unsigned)nblocks), dim3(DLAGP_FIN_THREADS), lds_bytes, stream,
                           ws.part.as<double>(), nparts, rows, (long)S, table, d_out, status,
                           plan ? plan->counts() : nullptr, src.pair_S ? d_out2 : nullptr, (long)src.pair_S, ev);
    } else {
        launch_finalize_general(stream, ws.part.as<const double>(), nparts, rows, (long)S, table, d_out, status,
                                plan ? plan->counts() : nullptr);
    }
    HIP_TRY(hipGetLastError());
    return DLAGP_OK;
}

// The launch group itself, or -- for a batch that mixes kernel flavours -- its homogeneous parts one after the other
// on the same stream, all raising the one non-PD flag of `g`.
static int launch_sweep(LaunchGroup &g, int64_t S, int kd, const SampleSource &src, double *d_out,
                        hipStream_t stream, TailRequest *tail = nullptr) {
    CallOptionsScope options_of_this_call;
    g_last_launch_groups = g.parts.empty() ? 1 : (int64_t)g.parts.size();
    int rc = enter_stream(g, stream);
    if (rc) return rc;
    if (g.parts.empty() || kd == 0 || S <= 0)  // kd == 0: null model, generic kernel anyway
        return launch_one(g, S, kd, src, d_out, stream, nullptr, false, nullptr, tail);
    if ((rc = grow(g, g.ws.status, sizeof(int)))) return rc;
    g.last_stream = stream;
    g.launched = true;
    bool first = true;
    for (auto &p : g.parts) {
        if ((rc = launch_one(*p, S, kd, src, d_out, stream, g.ws.status.as<int>(), !first))) return rc;
        first = false;
    }
    return DLAGP_OK;
}

// The host-pointer entry points (dlagp_eval, dlagp_null, dlagp_batch_*, dlagp_absorption) run on a stream of the CALLING
// THREAD, not on the null stream: two host threads that sweep different batches then overlap -- one's transfers and
// host-side work beside the other's kernels -- instead of queueing behind each other (the catalogue driver's two sweep
// threads).  Every such call still returns with its own work complete.
// Option "shared_work_stream" = 1 (the default since round 5) puts the KERNELS of all threads on one stream per device
// (transfers stay on the thread's stream, ordered against the kernels by events, and a call waits for its own completion
// event only): the kernels of different threads' calls then run strictly one after the other, each launch with the chip
// to itself, while uploads, downloads and host work overlap them.  In round 3 it made no difference on the catalogue
// driver (profiles/r03_experiments.txt item 15: the finalisation kernels between the accumulate launches left the
// streams little to contend for); with fused launches following each other, two threads' accumulate kernels on two
// streams interleave slot by slot and both lose their fill pattern: one stream is +21 % on a DR12Q-like catalogue, +7 % on
// the uniform mix (profiles/r05_experiments.txt item 12).  0: every calling thread launches on its own stream.
// Streams and staging memory of a host thread are BORROWED from process-wide free lists and handed back when the thread
// ends (thread_local holders' destructors, which make no HIP call: at process exit the runtime may already be gone).
// The catalogue driver starts five fresh worker threads per call; with per-thread streams and arenas that were never
// returned, a long-lived process lost two streams and >= 4 MB of page-locked memory per thread ever started.
struct StreamPool {
    std::mutex mutex;
    std::map<std::pair<int, int>, std::vector<hipStream_t>> free;  // (device, kind) -> idle streams
};
static StreamPool &stream_pool() {
    static StreamPool *pool = new StreamPool();  // never destroyed: thread_local destructors may run after statics'
    return *pool;
}
enum { STREAM_WORK = 0, STREAM_SETUP = 1 };

static hipStream_t borrow_stream(int device, int kind) {
    {
        StreamPool &pool = stream_pool();
        std::lock_guard<std::mutex> lock(pool.mutex);
        auto &idle = pool.free[{device, kind}];
        if (!idle.empty()) {
            hipStream_t s = idle.back();
            idle.pop_back();
            return s;
        }
    }
    hipStream_t s = nullptr;
    if (kind == STREAM_SETUP) {
        // highest priority: the set-up kernels are tiny, and behind a sweep's long-running waves they otherwise wait
        // for wave slots while a host thread sits in the synchronisation that ends every context creation
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
            hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest) == hipSuccess)
            return s;
    }
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;  // fall back to the null stream
    return s;
}

struct ThreadStreams {
    std::map<int, hipStream_t> streams;
    ~ThreadStreams() {
        StreamPool &pool = stream_pool();
        std::lock_guard<std::mutex> lock(pool.mutex);
        for (auto &kv : streams)
            if (kv.second) pool.free[{kv.first, STREAM_WORK}].push_back(kv.second);
    }
};

static hipStream_t work_stream(int device) {
    static thread_local ThreadStreams mine;
    auto it = mine.streams.find(device);
    if (it != mine.streams.end()) return it->second;
    hipStream_t s = borrow_stream(device, STREAM_WORK);
    mine.streams[device] = s;
    return s;
}

static std::mutex g_shared_stream_mutex;
static std::map<int, hipStream_t> g_shared_streams;  // the one kernel stream per device (option shared_work_stream)
static hipStream_t compute_stream(int device) {
    if (!g_opt_shared_work_stream) return work_stream(device);
    std::lock_guard<std::mutex> lock(g_shared_stream_mutex);
    auto it = g_shared_streams.find(device);
    if (it != g_shared_streams.end()) return it->second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
    g_shared_streams[device] = s;
    return s;
}
// ... without creating it: nullptr when no call has launched on the device yet
static hipStream_t shared_stream_if_any(int device) {
    std::lock_guard<std::mutex> lock(g_shared_stream_mutex);
    auto it = g_shared_streams.find(device);
    return it != g_shared_streams.end() ? it->second : nullptr;
}

// the calling thread's pair of events that order its transfer stream against the kernel stream within one call.  Like the
// streams they go back to a process-wide pool when the thread ends, so that a driver which starts a thread per chunk of
// a catalogue does not grow the number of events with the number of chunks.
struct CallEvents {
    hipEvent_t uploaded = nullptr, computed = nullptr;
    hipEvent_t gate = nullptr;  // a point inside the call's kernel sequence that a side launch waits for (made on first use)
};
struct EventPool {
    std::mutex mutex;
    std::map<int, std::vector<CallEvents>> free;
};
static EventPool &event_pool() {
    static EventPool *pool = new EventPool();  // never destroyed, for the same reason as the stream pool
    return *pool;
}
struct ThreadEvents {
    std::map<int, CallEvents> events;
    ~ThreadEvents() {
        EventPool &pool = event_pool();
        std::lock_guard<std::mutex> lock(pool.mutex);
        for (auto &kv : events)
            if (kv.second.uploaded && kv.second.computed) pool.free[kv.first].push_back(kv.second);
    }
};
static int call_events(int device, CallEvents **out) {
    static thread_local ThreadEvents mine;
    auto it = mine.events.find(device);
    if (it == mine.events.end()) {
        CallEvents ev;
        {
            EventPool &pool = event_pool();
            std::lock_guard<std::mutex> lock(pool.mutex);
            auto &idle = pool.free[device];
            if (!idle.empty()) {
                ev = idle.back();
                idle.pop_back();
            }
        }
        it = mine.events.emplace(device, ev).first;
    }
    CallEvents &ev = it->second;
    if (!ev.uploaded) HIP_TRY(hipEventCreateWithFlags(&ev.uploaded, hipEventDisableTiming));
    if (!ev.computed) HIP_TRY(hipEventCreateWithFlags(&ev.computed, hipEventDisableTiming));
    *out = &ev;
    return DLAGP_OK;
}

// launches copy_segments for `table` on `stream`: enough blocks for the largest piece, a row of them per piece
static void launch_copy_segments(const CopyTable &table, void *pack, hipStream_t stream) {
    size_t largest = 0;
    for (int i = 0; i < table.count; ++i) largest = std::max<size_t>(largest, (size_t)table.seg[i].bytes);
    const unsigned blocks = (unsigned)std::max<size_t>(
        1, std::min<size_t>((largest / 16 + DLAGP_SETUP_THREADS - 1) / DLAGP_SETUP_THREADS, 256));
    hipLaunchKernelGGL(copy_segments, dim3(blocks, (unsigned)table.count), dim3(DLAGP_SETUP_THREADS), 0, stream, table,
                       static_cast<unsigned char *>(pack));
}

static Option g_opt_pack_transfers = 1;
// One batch of host <-> device transfers through the workspace's pinned staging buffer.
//   add(): registers a transfer and returns nothing; run_uploads() copies host -> stage -> device (async on
//   the call's stream, the stage is not reused before the final synchronisation); run_downloads() enqueues
//   device -> stage, synchronises the device once and copies stage -> host.  Small transfers use hipMemcpy.
struct Transfers {
    struct Item {
        void *device;
        void *host;
        size_t bytes;
        size_t offset;
    };
    PinnedBuffer &stage;
    DeviceBuffer *pack;  // the workspace's device-side twin of `stage` (nullptr: every transfer a copy of its own)
    int device;
    hipStream_t compute, transfer;  // kernels of the call / copies of the call (may be the same stream)
    std::vector<Item> up, down;
    size_t total = 0;
    size_t base = 0;  // first byte of the staging buffer this call may use (a two-phase call's second phase: behind the first's)
    size_t up_bytes = 0;  // the uploads' stretch of the staging buffer: [base, base + up_bytes); the downloads follow
    bool packed = false;
    // Every transfer, small ones included, goes through the page-locked staging buffer as an ASYNCHRONOUS copy: a
    // blocking hipMemcpy of a few bytes costs 50-90 us of host time each, and a call made for ONE spectrum (the drop-in
    // classes: a dozen arrays in and out per model selection) spent 0.9 ms of its 1.4 ms in them
    // (profiles/r04_route_a_trace.txt).  One host wait per call, in run_downloads.
    // Round 6: a call's small transfers also cross the bus TOGETHER.  Each copy is a blit kernel of its own (4 us, one after
    // the other on the transfer stream: sixteen per spectrum on the drop-in route, all on the caller's critical path): up
    // to kPackLimit bytes in all, the pieces lie end to end in the staging buffer anyway, so one copy moves the stretch
    // to / from the workspace's pack buffer and copy_segments distributes / collects the pieces on the device.
    static constexpr size_t kDirect = 0;
    static constexpr size_t kPackLimit = (size_t)8 << 20;
    Transfers(Workspace &ws, int dev)
        : stage(ws.stage), pack(&ws.pack), device(dev), compute(compute_stream(dev)), transfer(work_stream(dev)) {}
    void upload(void *device_ptr, const void *host, size_t bytes) {
        if (bytes) up.push_back({device_ptr, const_cast<void *>(host), bytes, 0});
    }
    void download(void *host, const void *device_ptr, size_t bytes) {
        if (bytes && host) down.push_back({const_cast<void *>(device_ptr), host, bytes, 0});
    }
    // `also`: bytes a later phase of the same call will need behind this one's (both buffers are sized now: growing
    // them in the second phase would free memory the first phase's copies may still be reading)
    int reserve(size_t also = 0) {
        total = 0;
        for (auto *list : {&up, &down}) {
            for (Item &it : *list)
                if (it.bytes > kDirect) {
                    it.offset = base + total;
                    total += (it.bytes + 255) & ~(size_t)255;
                }
            if (list == &up) up_bytes = total;
        }
        packed = g_opt_pack_transfers && pack && up.size() + down.size() >= 3 && base + total + also <= kPackLimit &&
                 up.size() <= DLAGP_COPY_SEGMENTS && down.size() <= DLAGP_COPY_SEGMENTS;
        for (const Item &it : up) packed = packed && it.bytes % 4 == 0;
        for (const Item &it : down) packed = packed && it.bytes % 4 == 0;
        if (!(total + also)) return DLAGP_OK;
        const size_t want = base + total + also;
        // A buffer that is replaced must be idle.  Every call that used it ended with a wait for its copies -- unless it
        // failed half way (an allocation failure between its uploads and its downloads), on whichever thread's stream it
        // ran: growth of a buffer that has been used waits for the whole device first.  (Rare: the buffers only grow.
        // Found by the host-only sanitizer replay, round 6.)
        if ((stage.ptr && want > stage.bytes) || (packed && pack->ptr && want > pack->bytes)) HIP_TRY(hipDeviceSynchronize());
        int rc = stage.reserve(want);
        if (!rc && packed) rc = pack->reserve(want);
        return rc;
    }
    CopyTable table_of(const std::vector<Item> &items, int to_pack) const {
        CopyTable t{};
        t.count = (int)items.size();
        t.to_pack = to_pack;
        for (size_t i = 0; i < items.size(); ++i)
            t.seg[i] = CopySegment{(unsigned long long)(uintptr_t)items[i].device, (unsigned long long)items[i].offset,
                                   (unsigned long long)items[i].bytes};
        return t;
    }
    // host -> stage -> device on the transfer stream; the kernel stream waits for them (an event, no host wait)
    int run_uploads() {
        bool queued = false;
        for (const Item &it : up) {
            if (it.bytes > kDirect) {
                std::memcpy(static_cast<char *>(stage.ptr) + it.offset, it.host, it.bytes);
                if (!packed)
                    HIP_TRY(hipMemcpyAsync(it.device, static_cast<char *>(stage.ptr) + it.offset, it.bytes, hipMemcpyHostToDevice, transfer));
                queued = true;
            } else {
                HIP_TRY(hipMemcpy(it.device, it.host, it.bytes, hipMemcpyHostToDevice));
            }
        }
        if (queued && packed) {
            HIP_TRY(hipMemcpyAsync(static_cast<char *>(pack->ptr) + base, static_cast<char *>(stage.ptr) + base, up_bytes,
                                   hipMemcpyHostToDevice, transfer));
            launch_copy_segments(table_of(up, 0), pack->ptr, transfer);
            HIP_TRY(hipGetLastError());
        }
        if (queued && transfer != compute) {
            CallEvents *ev = nullptr;
            int rc = call_events(device, &ev);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ev->uploaded, transfer));
            HIP_TRY(hipStreamWaitEvent(compute, ev->uploaded, 0));
        }
        return DLAGP_OK;
    }
    // the call's kernels are complete on return (and nothing else is waited for: the kernel stream may already hold
    // another thread's launches behind them); device -> stage on the transfer stream, then stage -> host
    int run_downloads() {
        if (transfer != compute) {
            CallEvents *ev = nullptr;
            int rc = call_events(device, &ev);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ev->computed, compute));
            HIP_TRY(hipStreamWaitEvent(transfer, ev->computed, 0));
        }
        if (packed && !down.empty()) {
            launch_copy_segments(table_of(down, 1), pack->ptr, transfer);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(static_cast<char *>(stage.ptr) + base + up_bytes, static_cast<char *>(pack->ptr) + base + up_bytes,
                                   total - up_bytes, hipMemcpyDeviceToHost, transfer));
        } else {
            for (const Item &it : down)
                if (it.bytes > kDirect)
                    HIP_TRY(hipMemcpyAsync(static_cast<char *>(stage.ptr) + it.offset, it.device, it.bytes, hipMemcpyDeviceToHost, transfer));
        }
        HIP_TRY(hipStreamSynchronize(transfer));
        for (const Item &it : down) {
            if (it.bytes > kDirect) std::memcpy(it.host, static_cast<char *>(stage.ptr) + it.offset, it.bytes);
            else HIP_TRY(hipMemcpy(it.host, it.device, it.bytes, hipMemcpyDeviceToHost));
        }
        return DLAGP_OK;
    }
};

// The null model of every member (NullGP.log_model_evidence, reference null_gp.py:294-305) on `side`, beside whatever the
// group has on its kernel stream: what launch_one does for kd == 0 -- the generic kernel on one "sample" per sightline, four
// pixel ranges each, and its finalisation -- with partial sums and a non-PD flag of its own.  The same numbers, bit for bit.
static Option g_opt_null_side = 1;  // 0: the null model always as the call's last launch on the kernel stream
static Option g_opt_null_side_level = -2;  // the side launch waits for this level's evaluations (0-based; -1: the deepest level's, -2: the one before)
static int launch_null_side(LaunchGroup &g, double *d_out, hipStream_t side) {
    Workspace &ws = g.ws;
    const int nb = (int)g.ctxs.size();
    const int nparts = 4;
    const size_t nblocks = (size_t)nb;  // one block of 16 samples (one of them real) per sightline
    int rc;
    if (g.side_launched) {  // an earlier side launch still in flight (a call that failed or was cancelled before its downloads)
        HIP_TRY(hipStreamSynchronize(g.side_stream));
        g.side_launched = false;
    }
    if ((rc = ws.null_part.reserve((size_t)nparts * nblocks * g.row_width * 16 * sizeof(double)))) return rc;
    if ((rc = ws.null_status.reserve(sizeof(int)))) return rc;
    const SightlineDev *table = g.table();
    const int lds_entries = std::max(g.max_entries, g.max_npad);
    const int lds_pixels = g.max_npad;
    const size_t lds_bytes = (size_t)(lds_entries + lds_pixels + DLAGP_MAX_LINES + 1 + 8) * sizeof(double);
    if (lds_bytes > 160 * 1024)
        return fail(DLAGP_EINVAL, "sightline too long for the generic kernel: %zu B of LDS needed", lds_bytes);
    if (lds_bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(accumulate_generic),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    g.side_stream = side;
    g.side_launched = true;
    const SampleSource none{nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, 0, 0, 0, 0, 0};
    int *status = ws.null_status.as<int>();
    hipLaunchKernelGGL(accumulate_generic, dim3(1u, (unsigned)nb, (unsigned)nparts), dim3(DLAGP_GENERIC_THREADS), (uint32_t)lds_bytes,
                       side, table, 1L, 0, none, ws.null_part.as<double>(), status, lds_entries, lds_pixels, 0);
    HIP_TRY(hipGetLastError());
    if (g.k == DLAGP_FIN_K) {
        const bool row16 = (int64_t)nblocks <= g_opt_finalize_row16_blocks;
        auto fin = row16 ? finalize_rank20<true, false> : finalize_rank20<false, false>;
        hipLaunchKernelGGL(fin, dim3((unsigned)nblocks), dim3(DLAGP_FIN_THREADS), DLAGP_FINALIZE_LDS_DOUBLES * sizeof(double), side,
                           ws.null_part.as<double>(), nparts, (long)nb, 1L, table, d_out, status, (const unsigned char *)nullptr,
                           (double *)nullptr, 0L, EvidenceTail{});
    } else {
        launch_finalize_general(side, ws.null_part.as<const double>(), nparts, (long)nb, 1L, table, d_out, status, nullptr);
    }
    HIP_TRY(hipGetLastError());
    return DLAGP_OK;
}

// host-pointer wrapper: stage, launch, copy back, report non-PD as DLAGP_ELINALG
static int sweep_from_host(LaunchGroup &g, int64_t S, int kd, const double *z, const double *nhi, double *out) {
    Workspace &ws = g.ws;
    const long rows = (long)g.ctxs.size() * S;
    const size_t nsamp = (size_t)rows * (kd > 0 ? kd : 1);
    int rc;
    if ((rc = grow(g, ws.z, nsamp * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.nhi, nsamp * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.out, (size_t)rows * sizeof(double)))) return rc;
    Transfers tr(ws, g.device);
    const hipStream_t stream = tr.compute;
    if (kd > 0) {
        tr.upload(ws.z.ptr, z, nsamp * sizeof(double));
        tr.upload(ws.nhi.ptr, nhi, nsamp * sizeof(double));
    }
    tr.download(out, ws.out.ptr, (size_t)rows * sizeof(double));
    int status = 0;
    if ((rc = grow(g, ws.status, sizeof(int)))) return rc;  // the non-PD flag comes back with the results
    tr.download(&status, ws.status.ptr, sizeof(int));
    if ((rc = tr.reserve()) || (rc = tr.run_uploads())) return rc;
    rc = launch_sweep(g, S, kd, SampleSource{ws.z.as<double>(), ws.nhi.as<double>(), nullptr, 0, 0, 0, nullptr, nullptr, 0, 0, 0, 0, 0},
                      ws.out.as<double>(), stream);
    if (rc) return rc;
    if ((rc = tr.run_downloads())) return rc;
    g.launched = false;  // run_downloads drained the stream: nothing of this group is in flight (drain() has nothing to wait for)
    for (auto &p : g.parts) p->launched = false;
    if (status) return fail(DLAGP_ELINALG, "Matrix is not positive definite (I + M' D^-1 M)");
    return DLAGP_OK;
}

// The kernels that end an occupation level (kernels_common.h "level finish").  Spread form -- two launches that put one
// sightline's level on ~30 compute units -- whenever the cumulative weights fit a workgroup's LDS (8 S bytes beside 150
// static ones: S <= 18 000); the one-workgroup kernel otherwise, and when option "level_spread" is 0.  Same arithmetic, same
// bits either way.
static Option g_opt_level_spread = 1;
static Option g_opt_level_search_groups = 16;  // workgroups per sightline that share the searches of the spread form
static Option g_last_level_spread = 0;
// `also`: a second level finish of the same shape (S, sightlines) to run in the same launches -- the paired one-absorber
// model's beside the sweep's level 1 (two launches less per model selection); the one-workgroup forms run it separately.
static int launch_level_finish(int nb, hipStream_t stream, LevelArgs a, const LevelArgs *also = nullptr) {
    const long S = a.S;
    const size_t lds_bytes = (size_t)S * sizeof(double);
    const int lds_limit = 144000;  // beside the kernels' 16.7 KB of static LDS: S <= 18 000
    const bool in_lds = lds_bytes <= (size_t)lds_limit;
    const bool resamples = a.level + 1 < a.max_dlas;
    const bool spread = g_opt_level_spread != 0 && in_lds;
    g_last_level_spread = spread ? 1 : 0;
    // beyond the default limit (64 KB in all) the attribute is needed -- per device, so it is set at every such launch
    // rather than remembered (a process may drive several devices)
    const bool big_lds = resamples && lds_bytes > 46 * 1024;
    if (!spread && also) {
        int rc = launch_level_finish(nb, stream, *also);
        if (rc) return rc;
        also = nullptr;
    }
    if (spread && also) {
        const LevelArgsTwo both{a, *also, nb};
        hipLaunchKernelGGL(level_weights_two, dim3(DLAGP_LEVEL_LANES / 64, 2u * (unsigned)nb), dim3(64), 0, stream, both);
        if (big_lds)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(level_resample_two),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds_limit));
        const int groups = resamples ? (int)std::max<int64_t>(1, std::min<int64_t>(g_opt_level_search_groups, 64)) : 1;
        hipLaunchKernelGGL(level_resample_two, dim3((unsigned)groups, 2u * (unsigned)nb), dim3(DLAGP_LEVEL_SEARCH_THREADS),
                           resamples ? (uint32_t)lds_bytes : 0u, stream, both);
    } else if (spread) {
        hipLaunchKernelGGL(level_weights, dim3(DLAGP_LEVEL_LANES / 64, (unsigned)nb), dim3(64), 0, stream, a);
        if (big_lds)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(level_resample),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds_limit));
        const int groups = resamples ? (int)std::max<int64_t>(1, std::min<int64_t>(g_opt_level_search_groups, 64)) : 1;
        hipLaunchKernelGGL(level_resample, dim3((unsigned)groups, (unsigned)nb), dim3(DLAGP_LEVEL_SEARCH_THREADS),
                           resamples ? (uint32_t)lds_bytes : 0u, stream, a);
    } else if (in_lds) {
        if (big_lds)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(level_finish_fused<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds_limit));
        hipLaunchKernelGGL(level_finish_fused<true>, dim3((unsigned)nb), dim3(DLAGP_LEVEL_THREADS),
                           resamples ? (uint32_t)lds_bytes : 0u, stream, a);
    } else {
        hipLaunchKernelGGL(level_finish_fused<false>, dim3((unsigned)nb), dim3(DLAGP_LEVEL_THREADS), 0, stream, a);
    }
    HIP_TRY(hipGetLastError());
    return DLAGP_OK;
}

// The whole occupation sweep of reference dla_gp.py:92-225 for every sightline of the group, enqueued on
// `stream` with no host round trip between levels: per level one accumulate + finalize + level_finish.
// All pointers are device addresses; d_base may not be NULL when max_dlas > 1.
// `pair`: the one-absorber model that shares the sweep's redshift samples (the sub-DLA model beside the DLA model): its
// sweep rides in level 1's launch (pair launch, pair_eligible), its level is finished right behind it.
// `phase`: SWEEP_ALL; or the two halves of the two-phase call -- SWEEP_START: initialisation, level 1's launch, the paired
// model's evidence (none of which needs random numbers); SWEEP_FINISH: level 1's evidence and resampling, the deeper levels.
// `cache_decision`: START writes whether the profile cache is in use, FINISH reads it (the decision depends on the free
// memory of the moment).
struct PairSpec {
    const double *d_nhi;
    int64_t nhi_stride;
    double *d_evidences, *d_sample_ll;
};
enum { SWEEP_ALL = 0, SWEEP_START = 1, SWEEP_FINISH = 2 };
static int sweep_levels(LaunchGroup &g, int64_t S, int max_dlas, const double *d_sample_z, const double *d_sample_nhi,
                        int64_t nhi_stride, double min_z_separation, const double *d_uniforms, double *d_evidences,
                        double *d_sample_ll, int *d_base, hipStream_t stream, bool keep_sticky_flag = false,
                        const PairSpec *pair = nullptr, int phase = SWEEP_ALL, int *cache_decision = nullptr,
                        const std::function<int()> *after_level_launch = nullptr) {
    CallOptionsScope options_of_this_call;
    Workspace &ws = g.ws;
    const int nb = (int)g.ctxs.size();
    const size_t rows = (size_t)nb * S;
    int rc;
    if ((rc = enter_stream(g, stream))) return rc;
    if ((rc = grow(g, ws.level_ll, rows * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.cdf, rows * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.alive, (size_t)(nb + 1) * sizeof(int)))) return rc;  // [nb] = sticky non-PD flag
    if ((rc = grow(g, ws.level_x, (size_t)nb * DLAGP_LEVEL_EXCH * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.ties, (size_t)nb * std::max(max_dlas - 1, 1) * sizeof(int)))) return rc;
    if (pair && (rc = grow(g, ws.sub_level_ll, rows * sizeof(double)))) return rc;
    if (pair && (rc = grow(g, ws.sub_alive, (size_t)(nb + 1) * sizeof(int)))) return rc;
    if (pair && (rc = grow(g, ws.sub_cdf, rows * sizeof(double)))) return rc;
    if (pair && (rc = grow(g, ws.sub_level_x, (size_t)nb * DLAGP_LEVEL_EXCH * sizeof(double)))) return rc;
    const long n_ll = (long)rows * max_dlas, n_base = (long)rows * (max_dlas - 1);
    g.last_stream = stream;  // the fills below already use the group's scratch
    g.launched = true;
    if (phase != SWEEP_FINISH) {
        // a second sweep of the same call adds to the first one's non-PD flag (keep_sticky_flag)
        // (with it the paired model's own state: one level, its own alive flags -- the sticky flag is the sweep's)
        const long fill = std::max(std::max(n_ll, n_base), pair ? (long)rows : 0L);
        hipLaunchKernelGGL(sweep_init, dim3((unsigned)std::max<long>(1, std::min<long>((fill + 255) / 256, 2048))), dim3(256), 0, stream,
                           d_sample_ll, n_ll, d_evidences, (long)nb * max_dlas, ws.alive.as<int>(), (long)nb, keep_sticky_flag ? 0 : 1,
                           n_base > 0 ? d_base : nullptr, n_base, pair ? pair->d_sample_ll : nullptr, pair ? (long)rows : 0L,
                           pair ? pair->d_evidences : nullptr, pair ? (long)nb : 0L, pair ? ws.sub_alive.as<int>() : nullptr);
        HIP_TRY(hipGetLastError());
    }
    const SampleSource src0{d_sample_z, d_sample_nhi, d_base, 1, (long)nhi_stride, max_dlas - 1, nullptr, nullptr, 0, 0, 0, 0, 0};
    // the homogeneous launch groups of this sweep: the batch itself, or the parts of a batch of mixed flavours
    struct Target {
        LaunchGroup *g;
        SampleSource src;
        bool use_cache;
    };
    std::vector<Target> targets;
    if (g.parts.empty()) targets.push_back({&g, src0, false});
    else for (auto &p : g.parts) targets.push_back({p.get(), src0, false});
    g_last_launch_groups = (int64_t)targets.size();
    if (!g.parts.empty() && (rc = grow(g, ws.status, sizeof(int)))) return rc;  // the parts share this group's flag
    // Profile cache: every absorber of every level is one of the sightline's S QMC points, whose convolved
    // profile level 1 computes anyway.  Keeping them (8 S n_pad bytes per sightline) turns the Voigt work of
    // levels 2..max_dlas into HBM reads -- capacity traded for time on the shared FP64 pipe.
    bool all_cached = true;
    for (Target &t : targets) {
        LaunchGroup &tg = *t.g;
        Workspace &tws = tg.ws;
        // (ranks above 20 run every block once per column group: each would store / stream the rows again -- they recompute)
        bool use_cache = g_opt_profile_cache != 0 && max_dlas > 1 && tg.col_groups == 1;
        if (phase == SWEEP_FINISH && cache_decision) use_cache = use_cache && *cache_decision != 0;  // as START decided
        for (int level = 0; level < max_dlas; ++level) use_cache = use_cache && mfma_path_serves(tg, level + 1, level > 0);
        // tiled layout (dlagp_device.h): rows padded to whole tiles of 64 slots, points to whole groups of 16
        const size_t cache_row = ((size_t)tg.max_slots + DLAGP_CACHE_TILE_SLOTS - 1) / DLAGP_CACHE_TILE_SLOTS * DLAGP_CACHE_TILE_SLOTS;
        const size_t cache_points = ((size_t)S + DLAGP_MFMA_GROUP - 1) / DLAGP_MFMA_GROUP * DLAGP_MFMA_GROUP;
        // doubles per group of 16 points (padding between the groups -- 64 bytes to 12 KB, to spread the groups over the
        // memory channels -- changed nothing: profiles/r04_experiments.txt item 8)
        const size_t cache_group = DLAGP_MFMA_GROUP * cache_row;
        const size_t cache_sightline = cache_points / DLAGP_MFMA_GROUP * cache_group;
        const size_t cache_bytes = tg.ctxs.size() * cache_sightline * sizeof(double);
        const size_t want_bytes = cache_bytes;  // the profiles; the deeper levels multiply cached rows, nothing is written back
        // the cached levels address a sightline's rows with 32-bit byte offsets (raw buffer loads, kernels_mfma.h)
        use_cache = use_cache && cache_sightline * sizeof(double) < ((size_t)1 << 31);
        if (use_cache) {
            const size_t have = tws.cache.bytes;
            if (have < want_bytes) {
                size_t free_b = 0, total_b = 0;
                HIP_TRY(hipMemGetInfo(&free_b, &total_b));
                use_cache = want_bytes <= (size_t)g_opt_profile_cache_max_mb << 20 && want_bytes <= (free_b + have) / 2;
            }
            if (use_cache && (rc = grow(tg, tws.cache, cache_bytes))) return rc;
        }
        t.use_cache = use_cache;
        all_cached = all_cached && use_cache;
        if (use_cache) {
            t.src.cache = tws.cache.as<double>();
            t.src.cache_stride = (long)cache_row;
            t.src.cache_group_stride = (long)cache_group;
        }
    }
    g_last_sweep_cached = all_cached ? 1 : 0;
    if (cache_decision && phase != SWEEP_FINISH) *cache_decision = all_cached ? 1 : 0;
    g_last_sweep_paired = pair ? 1 : 0;
    const double log_s = std::log((double)S);
    // which level's launch the caller's hook follows: option null_side_level (0-based; negative: counted from the deepest,
    // -1 = the deepest), among the levels this phase launches
    const int first_launched = phase == SWEEP_FINISH ? 1 : 0;
    const int wanted = g_opt_null_side_level < 0 ? max_dlas + (int)g_opt_null_side_level : (int)g_opt_null_side_level;
    const int hook_level = std::min(std::max(wanted, first_launched), max_dlas - 1);
    for (int level = 0; level < max_dlas; ++level) {
        bool first = true;
        const bool launch = !(phase == SWEEP_FINISH && level == 0);  // FINISH: level 1 was launched by START
        for (Target &t : targets) {
            if (!launch) break;
            t.src.cache_mode = t.use_cache ? (level == 0 ? 1 : 2) : 0;
            if (pair && level == 0) {
                // both one-absorber models in one launch: rows 0..7 of every wave the paired model, rows 8..15 this sweep's
                SampleSource both = t.src;
                both.nhi = pair->d_nhi;
                both.nhi_stride = (long)pair->nhi_stride;
                both.nhi2 = d_sample_nhi;
                both.nhi2_stride = (long)nhi_stride;
                both.pair_S = (long)S;
                rc = launch_one(g, pair_virtual_samples(S), 1, both, ws.sub_level_ll.as<double>(), stream, nullptr, false,
                                ws.level_ll.as<double>());
            } else if (g.parts.empty()) rc = launch_one(g, S, level + 1, t.src, ws.level_ll.as<double>(), stream);
            else rc = launch_one(*t.g, S, level + 1, t.src, ws.level_ll.as<double>(), stream, ws.status.as<int>(), !first);
            if (rc) return rc;
            first = false;
        }
        // a level's evaluations are enqueued: what follows on this stream are small kernels (its finalisation and level
        // finish) -- the caller's moment for work that may run beside them
        if (launch && level == hook_level && after_level_launch && (rc = (*after_level_launch)())) return rc;
        if (phase == SWEEP_START) return DLAGP_OK;  // level 1's evidence and resampling wait for the uniforms -- and the paired
                                                    // model's level finish waits with them: it rides in the same launches
        LevelArgs this_level{ws.level_ll.as<const double>(), (long)S, level, max_dlas, src0, min_z_separation, log_s, d_uniforms,
                             d_sample_ll, d_evidences, d_base, ws.cdf.as<double>(), ws.level_x.as<double>(), ws.alive.as<int>(),
                             ws.status.as<const int>(), ws.alive.as<int>() + nb, ws.ties.as<int>()};
        if (pair && level == 0) {
            // the one-absorber model of the pair launch: one level, no resampling, its own weights / exchange / alive flags
            // (the launch's non-PD flag is the sweep's: this_level accumulates it)
            const SampleSource sub0{d_sample_z, pair->d_nhi, nullptr, 1, (long)pair->nhi_stride, 0, nullptr, nullptr, 0, 0, 0, 0, 0};
            const LevelArgs sub_level{ws.sub_level_ll.as<const double>(), (long)S, 0, 1, sub0, min_z_separation, log_s, nullptr,
                                      pair->d_sample_ll, pair->d_evidences, nullptr, ws.sub_cdf.as<double>(),
                                      ws.sub_level_x.as<double>(), ws.sub_alive.as<int>(), ws.status.as<const int>(),
                                      ws.alive.as<int>() + nb, nullptr};
            if ((rc = launch_level_finish(nb, stream, this_level, &sub_level))) return rc;
        } else if ((rc = launch_level_finish(nb, stream, this_level))) return rc;
    }
    return DLAGP_OK;
}

// Context creation runs on its own non-blocking stream (one per host thread and device), never on the null
// stream: a second host thread can then build the next batch's contexts while the first is inside a sweep --
// uploads and set-up kernels no longer queue behind the sweep's kernels, nor does hipDeviceSynchronize wait for
// them.
// (the stream and the page-locked arena travel together as a "lane": a thread borrows one per device, see above)
struct SetupLane {
    hipStream_t stream = nullptr;
    PinnedBuffer buf;  // Host -> device staging, see upload_staged
    size_t used = 0;
    bool dirty = false;  // handed back with copies possibly still reading the arena: the next borrower waits first
    // staged uploads not yet sent: they leave together (flush_uploads), one copy into `pack` and one kernel that moves the
    // pieces to their arrays
    DeviceBuffer pack;
    size_t pending_begin = 0;  // arena offset of the first byte not yet sent
    CopyTable pending{};
};
struct LanePool {
    std::mutex mutex;
    std::map<int, std::vector<SetupLane *>> free;
};
static LanePool &lane_pool() {
    static LanePool *pool = new LanePool();
    return *pool;
}
struct ThreadLanes {
    std::map<int, SetupLane *> lanes;
    ~ThreadLanes() {
        LanePool &pool = lane_pool();
        std::lock_guard<std::mutex> lock(pool.mutex);
        for (auto &kv : lanes) {
            kv.second->dirty = true;
            pool.free[kv.first].push_back(kv.second);
        }
    }
};
static SetupLane &setup_lane(int device) {
    static thread_local ThreadLanes mine;
    auto it = mine.lanes.find(device);
    if (it != mine.lanes.end()) return *it->second;
    SetupLane *lane = nullptr;
    {
        LanePool &pool = lane_pool();
        std::lock_guard<std::mutex> lock(pool.mutex);
        auto &idle = pool.free[device];
        if (!idle.empty()) {
            lane = idle.back();
            idle.pop_back();
        }
    }
    if (!lane) {
        lane = new SetupLane();
        lane->stream = borrow_stream(device, STREAM_SETUP);
    } else if (lane->dirty) {
        (void)hipStreamSynchronize(lane->stream);  // the previous owner's last copies have left the arena
        lane->used = 0;
        lane->pending_begin = 0;
        lane->pending.count = 0;
        lane->dirty = false;
    }
    mine.lanes[device] = lane;
    return *lane;
}
static hipStream_t setup_stream(int device) { return setup_lane(device).stream; }

// Host -> device on the set-up stream WITHOUT a synchronisation per array: the bytes are copied into a page-locked
// arena of the calling thread (so the source may be a temporary) and go out as one asynchronous copy each; the arena
// is recycled by setup_sync(), which every set-up sequence ends with.  (Context creation used to synchronise the
// stream eleven times -- once per array -- which was 0.6 ms per spectrum and the catalogue driver's bottleneck.)
typedef SetupLane SetupArena;
static SetupArena &setup_arena(int device) { return setup_lane(device); }

// Sends the calling thread's staged uploads: ONE copy of their stretch of the arena into the lane's pack buffer, one kernel
// that moves the pieces to their arrays.  Called in front of every set-up kernel that reads uploaded data, and by setup_sync.
static int flush_uploads(int device) {
    SetupLane &lane = setup_lane(device);
    if (!lane.pending.count) return DLAGP_OK;
    const size_t total = lane.used - lane.pending_begin;
    if (total > lane.pack.bytes) {
        // the previous flush's kernel may still be reading the pack buffer that a larger one replaces
        HIP_TRY(hipStreamSynchronize(lane.stream));
        int rc = lane.pack.reserve(std::max(total * 2, (size_t)1 << 20));
        if (rc) {
            lane.pending.count = 0;  // nothing of this stretch will be sent: the caller fails and releases its buffers
            lane.pending_begin = lane.used;
            return rc;
        }
    }
    lane.pending.to_pack = 0;
    const hipError_t err = hipMemcpyAsync(lane.pack.ptr, static_cast<char *>(lane.buf.ptr) + lane.pending_begin, total,
                                          hipMemcpyHostToDevice, lane.stream);
    if (err == hipSuccess) launch_copy_segments(lane.pending, lane.pack.ptr, lane.stream);
    lane.pending.count = 0;
    lane.pending_begin = lane.used;
    if (err != hipSuccess) return fail(DLAGP_EDEVICE, "staged upload failed: %s", hipGetErrorString(err));
    HIP_TRY(hipGetLastError());
    return DLAGP_OK;
}

// all staged uploads and kernels of the calling thread's set-up stream are complete on return
static int setup_sync(int device) {
    const int flushed = flush_uploads(device);
    HIP_TRY(hipStreamSynchronize(setup_stream(device)));
    SetupLane &lane = setup_lane(device);
    lane.used = 0;
    lane.pending_begin = 0;
    return flushed;
}

// A context created with DLAGP_FLAG_DEFER_SYNC carries an event recorded behind its set-up work on the CREATING thread's
// set-up stream.  Whoever uses the context first -- a batch built from it, possibly on another thread, an evaluation,
// dlagp_ctx_read_model, dlagp_destroy -- waits for that event; nothing rests on "the same thread synchronised its
// stream".
static int wait_setup(dlagp_ctx *ctx) {
    // the header promises "first use on ANY host thread": two first uses at once (two dlagp_batch_create calls sharing a
    // member) must not both synchronise and destroy the one event, nor release the scratch buffers twice
    std::lock_guard<std::mutex> lock(ctx->setup_mutex);
    if (!ctx->setup_pending) return DLAGP_OK;
    if (ctx->setup_event) {
        HIP_TRY(hipEventSynchronize(ctx->setup_event));
        (void)hipEventDestroy(ctx->setup_event);
        ctx->setup_event = nullptr;
    } else {
        HIP_TRY(hipDeviceSynchronize());  // no event could be recorded: the device-wide wait covers every stream
    }
    ctx->setup_pending = false;
    ctx->scratch_a.release();
    ctx->scratch_b.release();
    return DLAGP_OK;
}

// Staged uploads of a context's set-up: together (one copy + one copy_segments launch, flush_uploads) or each a copy of
// its own?  Measured on one box (round 6): together is 1.3 % of a one-spectrum model selection's wall time (nine
// dependent blits become two operations) -- and 8 % LESS throughput in a catalogue pipeline, where the set-up runs beside
// other batches' sweeps: the copies slip through while every kernel, copy_segments included, waits ~50 us for a wave slot
// among the accumulate waves, and the set-up's latency is on that pipeline's critical path (kernel stream dry at 5 % of the
// instants instead of 1.5 %).  So: 2 = together exactly when the device's kernel stream is idle at the moment (a caller that
// works one spectrum at a time), 1 = always, 0 = never.
static Option g_opt_pack_setup_uploads = 2;
static hipStream_t shared_stream_if_any(int device);
static bool pack_setup_uploads_now(int device) {
    const int64_t mode = g_opt_pack_setup_uploads;
    if (mode != 2) return mode != 0;
    hipStream_t shared = shared_stream_if_any(device);
    if (!g_opt_shared_work_stream || !shared) return true;
    const bool idle = hipStreamQuery(shared) == hipSuccess;
    (void)hipGetLastError();
    return idle;
}
// (bytes: a multiple of 4 -- doubles, ints and descriptor structs)
static int upload_staged(void *dst, const void *src, size_t bytes, int device) {
    if (!bytes) return DLAGP_OK;
    SetupArena &arena = setup_arena(device);
    const size_t need = (bytes + 255) & ~(size_t)255;
    int rc;
    if (arena.used + need > arena.buf.bytes) {
        if ((rc = setup_sync(device))) return rc;  // sends what is staged; copies in flight still read the old arena
        if (need > arena.buf.bytes && (rc = arena.buf.reserve(std::max(need * 2, (size_t)4 << 20)))) return rc;
    }
    if (arena.pending.count == DLAGP_COPY_SEGMENTS && (rc = flush_uploads(device))) return rc;
    char *slot = static_cast<char *>(arena.buf.ptr) + arena.used;
    std::memcpy(slot, src, bytes);
    if (!pack_setup_uploads_now(device)) {  // every array a copy of its own, sent at once
        if ((rc = flush_uploads(device))) return rc;
        arena.used += need;
        arena.pending_begin = arena.used;
        HIP_TRY(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, setup_stream(device)));
        return DLAGP_OK;
    }
    arena.pending.seg[arena.pending.count++] = CopySegment{(unsigned long long)(uintptr_t)dst,
                                                           (unsigned long long)(arena.used - arena.pending_begin),
                                                           (unsigned long long)((bytes + 3) & ~(size_t)3)};
    arena.used += need;
    return DLAGP_OK;
}

// fill the group-level summary from its members and upload the descriptor table; `positions`: where each member
// sits in the caller's arrays (a part of a heterogeneous batch), default 0, 1, 2, ...
// `table_staged`: the caller has staged the (one-member) table already, with the member's arrays (create_context).
static int finish_group(LaunchGroup &g, const std::vector<int> *positions = nullptr, bool defer_sync = false,
                        bool table_staged = false) {
    std::vector<SightlineDev> table;
    g.k = g.ctxs[0]->sl.k;
    g.row_width = g.ctxs[0]->sl.row_width;
    g.num_lines = g.ctxs[0]->sl.num_lines;
    g.broadening = g.ctxs[0]->sl.ntaps > 1;
    g.mfma_ok = true;
    g.grid_uniform = true;
    g.lls_break = g.ctxs[0]->sl.lls_break != 0;
    g.col_groups = g.ctxs[0]->col_groups;  // a batch's members share one rank (dlagp_batch_create)
    bool lls_mixed = false;
    for (dlagp_ctx *ctx : g.ctxs) {
        lls_mixed = lls_mixed || (ctx->sl.lls_break != 0) != g.lls_break;
        g.grid_uniform = g.grid_uniform && ctx->sl.grid_uniform;
        table.push_back(ctx->sl);
        table.back().batch_index = positions ? (*positions)[table.size() - 1] : (int)table.size() - 1;
        g.max_entries = std::max(g.max_entries, ctx->sl.num_entries);
        g.max_npad = std::max(g.max_npad, ctx->sl.n_pad);
        g.max_slots = std::max(g.max_slots, ctx->sl.n_slots);
        g.mfma_ok = g.mfma_ok && ctx->mfma_ok && ctx->sl.num_lines == g.num_lines &&
                    (ctx->sl.ntaps > 1) == g.broadening;
    }
    if (lls_mixed) g.mfma_ok = false;
    if (g.ctxs.size() == 1 && !positions && &g != &g.ctxs[0]->self && g.ctxs[0]->self.d_table.ptr) {
        g.table_alias = g.ctxs[0]->self.d_table.as<SightlineDev>();  // in place since the member's own set-up (wait_setup)
        return DLAGP_OK;
    }
    int rc = DLAGP_OK;
    if (!table_staged) {
        if ((rc = g.d_table.reserve(table.size() * sizeof(SightlineDev)))) return rc;
        // the table must be in place before the group's first launch, which runs on the caller's stream
        if ((rc = upload_staged(g.d_table.ptr, table.data(), table.size() * sizeof(SightlineDev), g.device))) return rc;
    }
    return defer_sync ? flush_uploads(g.device) : setup_sync(g.device);
}

// A batch whose members could each run the MFMA kernel but differ in line count, broadening or Lyman-limit flag:
// one homogeneous launch group per flavour (LaunchGroup::parts), members in the caller's order within each.
static int split_mixed_batch(LaunchGroup &g) {
    if (g.mfma_ok) return DLAGP_OK;
    for (const dlagp_ctx *ctx : g.ctxs)
        if (!ctx->mfma_ok) return DLAGP_OK;  // a member the MFMA kernel cannot serve (negative noise variance): the generic kernel serves the whole batch
    typedef std::tuple<int, int, int> Flavour;  // line count, taps, Lyman-limit flag
    std::map<Flavour, std::vector<int>> members;
    for (size_t b = 0; b < g.ctxs.size(); ++b) {
        const SightlineDev &sl = g.ctxs[b]->sl;
        members[Flavour(sl.num_lines, sl.ntaps, sl.lls_break)].push_back((int)b);
    }
    if (members.size() < 2) return DLAGP_OK;
    for (const auto &kv : members) {
        std::unique_ptr<LaunchGroup> part(new LaunchGroup());
        part->device = g.device;
        for (int b : kv.second) part->ctxs.push_back(g.ctxs[b]);
        int rc = finish_group(*part, &kv.second);
        if (rc) return rc;
        g.parts.push_back(std::move(part));
    }
    return DLAGP_OK;
}

// One-wave workgroups of the operand-building kernels for `elements` output values: one element per lane, 4096 waves at
// most.  (These kernels run beside other batches' sweeps, whose accumulate waves hold every register of their SIMD for
// ~200 us at a time, so a set-up wave starts only where an accumulate wave has just ended.  Fewer, fatter waves -- 16
// elements per lane -- were measured and dropped in round 6: each takes that much longer to find its slot AND to finish, the
// batch's contexts arrive late, and the kernel stream ran dry at 11.6 % of a catalogue run's instants instead of 1.8 %:
// 2140 instead of 2520 spectra/s.  The set-up's LATENCY is on the pipeline's critical path, not its wave count.)
static unsigned setup_blocks(size_t elements) {
    const size_t per_wave = (size_t)DLAGP_SETUP_THREADS;
    return (unsigned)std::max<size_t>(1, std::min<size_t>((elements + per_wave - 1) / per_wave, 4096));
}

// Everything of context creation that does not depend on where the model arrays come from:
// descriptor, window starts, pass planning, wavelength grid; allocates the per-pixel device arrays.
// `fill(ctx)` then has to leave y, mu, omega2, v (n_pad, padded 0/0/0/1) and dense_M ([n][sl.k]) on the device.
template <typename Fill>
static int create_context(dlagp_ctx **out, int n, int n_u, int k, const double *wavelengths, const uint8_t *keep,
                          int num_lines, int flags, int device, Fill fill) {
    const int broadening = flags & DLAGP_FLAG_BROADENING;
    const int lls_break = (flags & DLAGP_FLAG_LLS_BREAK) != 0;
    const bool defer_sync = (flags & DLAGP_FLAG_DEFER_SYNC) != 0;
    if (!out) return fail(DLAGP_EINVAL, "out is NULL");
    *out = nullptr;
    if (!wavelengths || !keep) return fail(DLAGP_EINVAL, "NULL input array");
    if (n <= 0 || n_u < n) return fail(DLAGP_EINVAL, "need 0 < n <= n_u (n=%d, n_u=%d)", n, n_u);
    if (k < 1 || k > DLAGP_MAX_RANK) return fail(DLAGP_EINVAL, "rank k=%d outside [1, %d]", k, DLAGP_MAX_RANK);
    if (num_lines < 1 || num_lines > DLAGP_MAX_LINES)
        return fail(DLAGP_EINVAL, "num_lines=%d outside [1, %d]", num_lines, DLAGP_MAX_LINES);
    if (flags & ~(DLAGP_FLAG_BROADENING | DLAGP_FLAG_LLS_BREAK | DLAGP_FLAG_DEFER_SYNC))
        return fail(DLAGP_EINVAL, "unknown flag bits 0x%x", flags);
    int kept = 0;
    for (int j = 0; j < n_u; ++j) kept += keep[j] != 0;
    if (kept != n)
        return fail(DLAGP_EINVAL, "keep flags select %d pixels but n=%d (len(absorption) == len(this_mu))", kept, n);
    int rc = ensure_device(device);
    if (rc) return rc;

    dlagp_ctx *ctx = new dlagp_ctx();
    ctx->device = device;
    ctx->user_k = k;
    // ranks up to 20 run as rank 20: zero columns of M add an identity block to B and nothing to u,
    // so the density is unchanged and one MFMA tile layout serves them all
    const int k_eff = k <= DLAGP_MFMA_RANK ? DLAGP_MFMA_RANK : k;
    SightlineDev &sl = ctx->sl;
    sl.n = n;
    sl.n_pad = (n + 7) / 8 * 8;
    sl.nsteps = sl.n_pad / DLAGP_STEP_PIXELS;
    sl.k = k_eff;
    sl.lls_break = lls_break;
    sl.ntaps = broadening ? DLAGP_NUM_TAPS : 1;
    sl.num_entries = broadening ? n_u + DLAGP_NUM_TAPS - 1 : n_u;
    sl.num_lines = num_lines;
    sl.ntw = (k_eff * (k_eff + 1) / 2 + DLAGP_TILE_COLS - 1) / DLAGP_TILE_COLS;
    sl.ntq = (k_eff + DLAGP_TILE_COLS - 1) / DLAGP_TILE_COLS;
    sl.row_width = (sl.ntw + sl.ntq) * DLAGP_TILE_COLS + DLAGP_ROW_EXTRA;
    sl.npair = (sl.ntw + sl.ntq + 1) / 2;
    for (int t = 0; t < DLAGP_NUM_TAPS; ++t) sl.taps[t] = broadening ? DLAGP_INSTRUMENT_PROFILE[t] : (t == 0 ? 1.0 : 0.0);
    static_assert(DLAGP_NUM_TAPS == 7, "the MFMA kernel folds the symmetric 7-tap profile into 4 coefficients");
    for (int t = 0; t < 3; ++t)
        if (DLAGP_INSTRUMENT_PROFILE[t] != DLAGP_INSTRUMENT_PROFILE[6 - t]) {
            delete ctx;
            return fail(DLAGP_EINVAL, "instrument profile is not symmetric");
        }

    // window start of kept pixel i = index of its unmasked pixel: with broadening the padded grid
    // has 3 extra entries in front, so the 7-tap "valid" window of unmasked pixel j is j .. j+6
    std::vector<int> &hpos = ctx->h_pos;
    hpos.assign(sl.n_pad, 0);
    for (int j = 0, i = 0; j < n_u; ++j)
        if (keep[j]) hpos[i++] = j;
    for (int i = n; i < sl.n_pad; ++i) hpos[i] = hpos[n - 1];
    {
        // finite, increasing, uniform in log10 to 0.4 entry?  (core patch of the MFMA kernel, kernels_mfma.h)
        const int ne = sl.num_entries;
        bool uniform = ne >= 2 && wavelengths[0] > 0.0 && std::isfinite(wavelengths[0]) && std::isfinite(wavelengths[ne - 1]);
        const double log0 = uniform ? std::log10(wavelengths[0]) : 0.0;
        const double dlog = uniform ? (std::log10(wavelengths[ne - 1]) - log0) / (ne - 1) : 0.0;
        uniform = uniform && dlog > 0.0;
        // entry e within 0.4 entries of log0 + e dlog, without a logarithm per entry (1200 of them were a third of this
        // function's host time): the bounds w0 10^((e -+ 0.4) dlog) advance by one multiplication per entry; their rounding
        // drift over a grid (~1e-13 relative) is nine orders below the tolerance (0.4 dlog ln 10 ~ 1e-4 relative)
        if (uniform) {
            const double ratio = std::pow(10.0, dlog);
            double lo = wavelengths[0] * std::pow(10.0, -0.4 * dlog), hi = wavelengths[0] * std::pow(10.0, 0.4 * dlog);
            for (int e = 1; uniform && e < ne; ++e) {
                lo *= ratio;
                hi *= ratio;
                uniform = wavelengths[e] > wavelengths[e - 1] && wavelengths[e] >= lo && wavelengths[e] <= hi;
            }
        }
        sl.grid_uniform = uniform ? 1 : 0;
        sl.grid_log0 = log0;
        sl.grid_inv_dlog = uniform ? 1.0 / dlog : 0.0;
        // the core half-width must stay below 3.1 entries for the 7-candidate window: 16 Doppler widths are
        // 16 sqrt2 sigma / c ln10 = 2.98e-4 dex
        if (uniform && 2.98e-4 / dlog > 3.1) sl.grid_uniform = 0;
    }
    // MFMA path: slot order (zero-weight slots close a chunk at a wide mask gap, so every mask is plannable)
    std::vector<int> slot_pixel;
    // Every rank runs the MFMA kernel: up to 20 as rank 20 (one column group); above, the k (k + 1) / 2 + k columns in
    // groups of 210 + 20, each contracted by its own waves (kernels_mfma.h, kernels_common.h regroup_native)
    ctx->mfma_ok = true;
    ctx->col_groups = k_eff == DLAGP_MFMA_RANK ? 1
                      : std::max((k_eff * (k_eff + 1) / 2 + DLAGP_MFMA_GROUP_W_COLS - 1) / DLAGP_MFMA_GROUP_W_COLS,
                                 (k_eff + DLAGP_MFMA_GROUP_Q_COLS - 1) / DLAGP_MFMA_GROUP_Q_COLS);
    if (ctx->mfma_ok) {
        std::vector<int> real_pos(hpos.begin(), hpos.begin() + n);
        plan::plan_slots(real_pos, n, sl.ntaps, slot_pixel, ctx->h_slot_pos);
    }
    sl.n_slots = (int)slot_pixel.size();
    sl.nsteps_m = sl.n_slots / DLAGP_STEP_PIXELS;

    const size_t px = (size_t)sl.n_pad * sizeof(double);
    const size_t p_bytes = (size_t)sl.nsteps * sl.npair * 2 * DLAGP_TILE_DOUBLES * sizeof(double);
    const size_t pm_bytes = (size_t)ctx->col_groups * sl.nsteps_m * DLAGP_MFMA_PAIRS * 2 * DLAGP_TILE_DOUBLES * sizeof(double);
    rc = ctx->y.reserve(px);
    if (!rc) rc = ctx->mu.reserve(px);
    if (!rc) rc = ctx->omega2.reserve(px);
    if (!rc) rc = ctx->v.reserve(px);
    if (!rc) rc = ctx->pos.reserve((size_t)sl.n_pad * sizeof(int));
    if (!rc) rc = ctx->lam.reserve((size_t)sl.num_entries * sizeof(double));
    if (!rc) rc = ctx->dense_M.reserve((size_t)n * k_eff * sizeof(double));
    if (!rc) rc = ctx->P.reserve(p_bytes);
    const hipStream_t setup = setup_stream(device);
    if (!rc && ctx->mfma_ok) {
        const size_t sb = (size_t)sl.n_slots * sizeof(int);
        rc = ctx->aux.reserve((size_t)sl.n_slots * DLAGP_AUX_FIELDS * sizeof(double));
        if (!rc) rc = ctx->Pm.reserve(pm_bytes);
        if (!rc) rc = ctx->slot_pixel.reserve(sb);
        if (!rc) rc = ctx->slot_pos.reserve(sb);
        if (!rc) rc = upload_staged(ctx->slot_pixel.ptr, slot_pixel.data(), sb, device);
        if (!rc) rc = upload_staged(ctx->slot_pos.ptr, ctx->h_slot_pos.data(), sb, device);
    }
    sl.y = ctx->y.as<double>();
    sl.mu = ctx->mu.as<double>();
    sl.omega2 = ctx->omega2.as<double>();
    sl.v = ctx->v.as<double>();
    sl.pos = ctx->pos.as<int>();
    sl.lam = ctx->lam.as<double>();
    sl.P = ctx->P.as<double>();
    sl.Pm = ctx->Pm.as<double>();
    sl.aux = ctx->aux.as<double>();
    // the descriptor is complete: the table of the context's own group {this} travels with the arrays (one copy, one
    // copy_segments launch for all of them -- staged at the end of the sequence it was a second pair, 10 us of every
    // one-spectrum call)
    if (!rc) rc = ctx->self.d_table.reserve(sizeof(SightlineDev));
    if (!rc) {
        SightlineDev own = sl;
        own.batch_index = 0;
        rc = upload_staged(ctx->self.d_table.ptr, &own, sizeof(SightlineDev), device);
    }
    if (!rc) rc = upload_staged(ctx->pos.ptr, hpos.data(), (size_t)sl.n_pad * sizeof(int), device);
    if (!rc) rc = upload_staged(ctx->lam.ptr, wavelengths, (size_t)sl.num_entries * sizeof(double), device);
    if (!rc) rc = fill(ctx);
    if (rc) {
        (void)setup_sync(device);  // staged copies may still be writing the buffers that go back to the pool
        dlagp_destroy(ctx);
        return rc;
    }
    // Invalid input the reference answers with NaN or LinAlgError: with a negative noise variance d = omega2 a^2 + v can
    // be negative, and np.log(d) (null_gp.py:356) is NaN for ANY negative pixel.  The MFMA kernel's running product of
    // d would turn an even number of negative pixels into a finite number, so such a sightline takes the generic kernel,
    // which takes the logarithm pixel by pixel as the reference does.
    if (ctx->negative_noise && !g_opt_negative_noise_mfma) ctx->mfma_ok = false;

    if ((rc = flush_uploads(device))) {  // whatever `fill` and the tables above have staged: the kernels below read it
        (void)setup_sync(device);
        dlagp_destroy(ctx);
        return rc;
    }
    if (ctx->mfma_ok) {
        hipLaunchKernelGGL(fill_records, dim3((sl.n_slots + DLAGP_SETUP_THREADS - 1) / DLAGP_SETUP_THREADS), dim3(DLAGP_SETUP_THREADS), 0, setup, sl.n_slots,
                           ctx->slot_pixel.as<const int>(), ctx->slot_pos.as<const int>(), sl.y, sl.mu, sl.omega2, sl.v,
                           ctx->aux.as<double>());
        hipLaunchKernelGGL(build_mfma_operand, dim3(setup_blocks(pm_bytes / sizeof(double))), dim3(DLAGP_SETUP_THREADS), 0, setup,
                           ctx->dense_M.as<double>(), ctx->slot_pixel.as<const int>(), sl.n_slots, ctx->Pm.as<double>(), k_eff,
                           ctx->col_groups);
    }
    hipLaunchKernelGGL(build_shared_operand, dim3(setup_blocks(p_bytes / sizeof(double))), dim3(DLAGP_SETUP_THREADS), 0, setup,
                       ctx->dense_M.as<double>(), n, sl.n_pad, k_eff, sl.ntw, sl.ntq, sl.npair, ctx->P.as<double>());
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        (void)setup_sync(device);
        dlagp_destroy(ctx);
        return fail(DLAGP_EDEVICE, "context set-up kernels failed: %s", hipGetErrorString(err));
    }
    ctx->self.device = device;
    ctx->self.ctxs.push_back(ctx);
    // ends with the one synchronisation of the whole set-up sequence -- unless the caller defers it to the batch it is
    // building (DLAGP_FLAG_DEFER_SYNC); the kernels' temporaries then stay with the context until it is destroyed
    rc = finish_group(ctx->self, nullptr, defer_sync, /*table_staged=*/true);
    ctx->setup_pending = defer_sync;
    if (defer_sync && !rc) {
        if (hipEventCreateWithFlags(&ctx->setup_event, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(ctx->setup_event, setup) != hipSuccess) {
            if (ctx->setup_event) (void)hipEventDestroy(ctx->setup_event);
            ctx->setup_event = nullptr;  // wait_setup falls back to a device-wide wait
        }
    }
    if (rc) {
        // finish_group failed before its synchronisation (the descriptor table could not be allocated): the set-up kernels
        // may still be reading the scratch inputs, which dlagp_destroy hands back to the pool -- wait first.  (Found by the
        // host-only sanitizer replay with injected allocation failures, round 5: the scratch used to be released above this.)
        (void)setup_sync(device);
        dlagp_destroy(ctx);
        return rc;
    }
    if (!defer_sync) {  // finish_group has synchronised the set-up stream: the kernels' inputs are idle
        ctx->scratch_a.release();
        ctx->scratch_b.release();
    }
    *out = ctx;
    return DLAGP_OK;
}

// upload n doubles padded to n_pad with `pad`
static int upload_padded(DeviceBuffer &buf, const double *src, int n, int n_pad, double pad, int device) {
    std::vector<double> host(n_pad, pad);
    std::copy(src, src + n, host.begin());
    return upload_staged(buf.ptr, host.data(), (size_t)n_pad * sizeof(double), device);
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int dlagp_abi_version(void) { return 3; }  // 2: DLAGP_ENOMEM, exception barrier; 3: near-tie counts of the resampling

int dlagp_device_count(void) try {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
} DLAGP_CATCH_INT("dlagp_device_count")

const char *dlagp_last_error(void) { return g_last_error.c_str(); }

int dlagp_create(dlagp_ctx **out, int n, int n_u, int k, const double *y, const double *mu, const double *M,
                 ptrdiff_t m_row_stride, ptrdiff_t m_col_stride, const double *omega2, const double *v,
                 const double *wavelengths, const uint8_t *keep, int num_lines, int broadening, int device) try {
    if (out) *out = nullptr;
    if (!y || !mu || !M || !omega2 || !v) return fail(DLAGP_EINVAL, "NULL input array");
    return create_context(out, n, n_u, k, wavelengths, keep, num_lines, broadening, device, [&](dlagp_ctx *ctx) -> int {
        const SightlineDev &sl = ctx->sl;
        for (int i = 0; i < n; ++i) ctx->negative_noise = ctx->negative_noise || omega2[i] < 0.0 || v[i] < 0.0;
        int rc = upload_padded(ctx->y, y, n, sl.n_pad, 0.0, ctx->device);
        if (!rc) rc = upload_padded(ctx->mu, mu, n, sl.n_pad, 0.0, ctx->device);
        if (!rc) rc = upload_padded(ctx->omega2, omega2, n, sl.n_pad, 0.0, ctx->device);
        if (!rc) rc = upload_padded(ctx->v, v, n, sl.n_pad, 1.0, ctx->device);
        if (rc) return rc;
        std::vector<double> hM((size_t)n * sl.k, 0.0);
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < k; ++c) hM[(size_t)i * sl.k + c] = M[i * m_row_stride + c * m_col_stride];
        return upload_staged(ctx->dense_M.ptr, hM.data(), hM.size() * sizeof(double), ctx->device);
    });
} DLAGP_CATCH_INT("dlagp_create")

int dlagp_model_create(dlagp_model **out, int n_grid, int k, const double *rest_wavelengths, const double *mu,
                       const double *M, ptrdiff_t m_row_stride, ptrdiff_t m_col_stride, const double *log_omega,
                       double log_c_0, double log_tau_0, double log_beta, double prev_tau_0, double prev_beta,
                       int num_forest_lines, int device) try {
    if (!out) return fail(DLAGP_EINVAL, "out is NULL");
    *out = nullptr;
    if (!rest_wavelengths || !mu || !M || !log_omega) return fail(DLAGP_EINVAL, "NULL input array");
    if (n_grid < 2) return fail(DLAGP_EINVAL, "the model grid needs at least two points");
    if (k < 1 || k > DLAGP_MAX_RANK) return fail(DLAGP_EINVAL, "rank k=%d outside [1, %d]", k, DLAGP_MAX_RANK);
    if (num_forest_lines < 1 || num_forest_lines > DLAGP_MAX_LINES)
        return fail(DLAGP_EINVAL, "num_forest_lines=%d outside [1, %d]", num_forest_lines, DLAGP_MAX_LINES);
    for (int i = 1; i < n_grid; ++i)
        if (!(rest_wavelengths[i] > rest_wavelengths[i - 1]))
            return fail(DLAGP_EINVAL, "rest_wavelengths must be strictly increasing");
    int rc = ensure_device(device);
    if (rc) return rc;
    dlagp_model *m = new dlagp_model();
    m->device = device;
    m->grid_min = rest_wavelengths[0];
    m->grid_max = rest_wavelengths[n_grid - 1];
    std::vector<double> cols((size_t)k * n_grid);
    for (int c = 0; c < k; ++c)
        for (int i = 0; i < n_grid; ++i) cols[(size_t)c * n_grid + i] = M[i * m_row_stride + c * m_col_stride];
    const size_t gb = (size_t)n_grid * sizeof(double);
    rc = m->grid.reserve(gb);
    if (!rc) rc = m->mu.reserve(gb);
    if (!rc) rc = m->log_omega.reserve(gb);
    if (!rc) rc = m->M.reserve(cols.size() * sizeof(double));
    hipError_t err = hipSuccess;
    if (!rc) err = hipMemcpy(m->grid.ptr, rest_wavelengths, gb, hipMemcpyHostToDevice);
    if (!rc && err == hipSuccess) err = hipMemcpy(m->mu.ptr, mu, gb, hipMemcpyHostToDevice);
    if (!rc && err == hipSuccess) err = hipMemcpy(m->log_omega.ptr, log_omega, gb, hipMemcpyHostToDevice);
    if (!rc && err == hipSuccess) err = hipMemcpy(m->M.ptr, cols.data(), cols.size() * sizeof(double), hipMemcpyHostToDevice);
    if (!rc && err != hipSuccess) rc = fail(DLAGP_EDEVICE, "model upload failed: %s", hipGetErrorString(err));
    if (rc) {
        dlagp_model_destroy(m);
        return rc;
    }
    ModelDev &d = m->dev;
    d.n_grid = n_grid;
    d.k = k;
    d.num_forest_lines = num_forest_lines;
    d.grid = m->grid.as<double>();
    d.mu = m->mu.as<double>();
    d.log_omega = m->log_omega.as<double>();
    d.M = m->M.as<double>();
    d.prev_beta = prev_beta;
    d.beta = std::exp(log_beta);
    d.c_0 = std::exp(log_c_0);
    const double tau_0 = std::exp(log_tau_0);
    // effective_optical_depth.py:60-77: lines in Angstrom; scale = tau_0 * f / f[0] * lines / lines[0], left to right
    const double line0 = DLAGP_TRANSITION_WAVELENGTHS_CM[0] * 1e8;
    for (int l = 0; l < DLAGP_MAX_LINES; ++l) {
        const double line = DLAGP_TRANSITION_WAVELENGTHS_CM[l] * 1e8;
        const double f = DLAGP_OSCILLATOR_STRENGTHS[l], f0 = DLAGP_OSCILLATOR_STRENGTHS[0];
        d.lines[l] = line;
        d.scale_prev[l] = prev_tau_0 * f / f0 * line / line0;
        d.scale_noise[l] = tau_0 * f / f0 * line / line0;
    }
    *out = m;
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_model_create")

void dlagp_model_destroy(dlagp_model *model) try {
    if (!model) return;
    (void)hipSetDevice(model->device);
    model->grid.release();
    model->mu.release();
    model->log_omega.release();
    model->M.release();
    delete model;
} DLAGP_CATCH_VOID("dlagp_model_destroy")

int dlagp_setup_sync(int device) try {
    int rc = ensure_device(device);
    return rc ? rc : setup_sync(device);
} DLAGP_CATCH_INT("dlagp_setup_sync")

// NullGP.get_interp on the device for a context under construction (the `fill` of create_context): y and v go up, mu,
// omega2 and the factor are interpolated from the model at the kept pixels' rest wavelengths x and suppressed by the
// Lyman-series mean flux at the observed this_wavelengths (reference null_gp.py:179-242)
static int check_model_range(const dlagp_model *model, int n, const double *x) {
    for (int i = 0; i < n; ++i)  // scipy's interp1d raises ValueError outside the grid (bounds_error)
        if (!(x[i] >= model->grid_min && x[i] <= model->grid_max))
            return fail(DLAGP_EINVAL, "rest wavelength %g outside the model grid [%g, %g]", x[i], model->grid_min,
                        model->grid_max);
    return DLAGP_OK;
}
static int fill_from_model(dlagp_ctx *ctx, const dlagp_model *model, int n, const double *x, const double *y, const double *v,
                           const double *this_wavelengths, double z_qso) {
    const SightlineDev &sl = ctx->sl;
    const hipStream_t setup = setup_stream(ctx->device);
    for (int i = 0; i < n; ++i) ctx->negative_noise = ctx->negative_noise || v[i] < 0.0;  // omega2 = exp(..) >= 0
    int rc = upload_padded(ctx->y, y, n, sl.n_pad, 0.0, ctx->device);
    if (!rc) rc = upload_padded(ctx->v, v, n, sl.n_pad, 1.0, ctx->device);
    if (rc) return rc;
    // the kernel's inputs live until the end of the set-up sequence (create_context releases them after its one
    // synchronisation: a buffer handed back to the pool while the kernel is in flight could be given to another thread)
    DeviceBuffer &dx = ctx->scratch_a, &dw = ctx->scratch_b;
    rc = dx.reserve((size_t)n * sizeof(double));
    if (!rc) rc = dw.reserve((size_t)n * sizeof(double));
    if (!rc) rc = upload_staged(dx.ptr, x, (size_t)n * sizeof(double), ctx->device);
    if (!rc) rc = upload_staged(dw.ptr, this_wavelengths, (size_t)n * sizeof(double), ctx->device);
    if (!rc) rc = flush_uploads(ctx->device);  // the context's arrays staged so far leave as one copy; interp_model reads x and w
    if (rc) return rc;
    hipLaunchKernelGGL(interp_model, dim3((sl.n_pad + DLAGP_INTERP_PIXELS - 1) / DLAGP_INTERP_PIXELS),
                       dim3(DLAGP_INTERP_THREADS), 0, setup, model->dev, n, sl.n_pad,
                       sl.k, dx.as<const double>(), dw.as<const double>(), z_qso, ctx->mu.as<double>(),
                       ctx->omega2.as<double>(), ctx->dense_M.as<double>());
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(DLAGP_EDEVICE, "interp_model failed: %s", hipGetErrorString(err));
    return DLAGP_OK;
}

int dlagp_create_from_model(dlagp_ctx **out, const dlagp_model *model, int n, int n_u, const double *x,
                            const double *y, const double *v, const double *this_wavelengths, double z_qso,
                            const double *wavelengths, const uint8_t *keep, int num_lines, int broadening) try {
    if (out) *out = nullptr;
    if (!model) return fail(DLAGP_EINVAL, "model is NULL");
    if (!x || !y || !v || !this_wavelengths) return fail(DLAGP_EINVAL, "NULL input array");
    if (int rc = check_model_range(model, n, x)) return rc;
    return create_context(out, n, n_u, model->dev.k, wavelengths, keep, num_lines, broadening, model->device,
                          [&](dlagp_ctx *ctx) -> int { return fill_from_model(ctx, model, n, x, y, v, this_wavelengths, z_qso); });
} DLAGP_CATCH_INT("dlagp_create_from_model")

int dlagp_ctx_read_model(dlagp_ctx *ctx, double *this_mu, double *this_M, double *this_omega2) try {
    if (!ctx) return fail(DLAGP_EINVAL, "context is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = wait_setup(ctx)) return rc;
    const int n = ctx->sl.n, k = ctx->user_k, ks = ctx->sl.k;
    if (this_mu) HIP_TRY(hipMemcpy(this_mu, ctx->mu.ptr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (this_omega2) HIP_TRY(hipMemcpy(this_omega2, ctx->omega2.ptr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (this_M) {
        std::vector<double> host((size_t)n * ks);
        HIP_TRY(hipMemcpy(host.data(), ctx->dense_M.ptr, host.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < k; ++c) this_M[(size_t)i * k + c] = host[(size_t)i * ks + c];
    }
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_ctx_read_model")

void dlagp_destroy(dlagp_ctx *ctx) try {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    // deferred set-up (DLAGP_FLAG_DEFER_SYNC) that nobody waited for: its copies and kernels may still be writing the
    // buffers released below
    (void)wait_setup(ctx);
    if (ctx->setup_event) (void)hipEventDestroy(ctx->setup_event);
    {
        // batches are meant to be destroyed before their members; if one is still alive, its launches may be using
        // this context's arrays
        std::lock_guard<std::mutex> lock(g_membership_mutex);
        for (dlagp_batch *batch : ctx->batches) batch->group.drain();
        for (dlagp_batch *batch : ctx->batches) {  // and it must not touch the context again
            std::replace(batch->group.ctxs.begin(), batch->group.ctxs.end(), ctx, (dlagp_ctx *)nullptr);
            if (batch->group.table_alias == ctx->self.table()) batch->group.table_alias = nullptr;
            for (auto &part : batch->group.parts)
                std::replace(part->ctxs.begin(), part->ctxs.end(), ctx, (dlagp_ctx *)nullptr);
        }
    }
    ctx->self.release();  // drains the stream of the last asynchronous launch
    ctx->y.release();
    ctx->mu.release();
    ctx->omega2.release();
    ctx->v.release();
    ctx->pos.release();
    ctx->lam.release();
    ctx->P.release();
    ctx->aux.release();
    ctx->Pm.release();
    ctx->slot_pixel.release();
    ctx->slot_pos.release();
    ctx->dense_M.release();
    ctx->scratch_a.release();
    ctx->scratch_b.release();
    delete ctx;
} DLAGP_CATCH_VOID("dlagp_destroy")

int dlagp_ctx_n(const dlagp_ctx *ctx) { return ctx ? ctx->sl.n : DLAGP_EINVAL; }
int dlagp_ctx_k(const dlagp_ctx *ctx) { return ctx ? ctx->user_k : DLAGP_EINVAL; }
int dlagp_ctx_device(const dlagp_ctx *ctx) { return ctx ? ctx->device : DLAGP_EINVAL; }

// a batch one of whose members has been destroyed can no longer launch
static int batch_alive(const dlagp_batch *batch) {
    if (!batch) return fail(DLAGP_EINVAL, "NULL batch");
    for (const dlagp_ctx *ctx : batch->group.ctxs)
        if (!ctx) return fail(DLAGP_EINVAL, "a member context of this batch has been destroyed");
    return DLAGP_OK;
}

// dlagp_batch_model_evidences_start has run and _finish has not, and another call arrives on the batch: that call reuses
// the staging memory and the sample tables the pending phase left for _finish, so the pending call is over -- its
// kernels drain, and a later _finish reports "without a matching _start" instead of sweeping over overwritten inputs.
static void cancel_pending(dlagp_batch *batch) {
    if (!batch || !batch->pending.active) return;
    batch->group.drain();
    batch->pending.active = false;
}

static int check_eval_args(const void *ctx, int64_t S, int kd, const void *z, const void *nhi, const void *out) {
    if (!ctx) return fail(DLAGP_EINVAL, "context is NULL");
    if (S < 0) return fail(DLAGP_EINVAL, "S=%lld is negative", (long long)S);
    if (kd < 1 || kd > DLAGP_MAX_ABSORBERS)
        return fail(DLAGP_EINVAL, "kd=%d outside [1, %d]", kd, DLAGP_MAX_ABSORBERS);
    if (S > 0 && (!z || !nhi || !out)) return fail(DLAGP_EINVAL, "NULL sample or output array");
    return DLAGP_OK;
}

int dlagp_eval(dlagp_ctx *ctx, int64_t S, int kd, const double *z, const double *nhi, double *out) try {
    int rc = check_eval_args(ctx, S, kd, z, nhi, out);
    if (rc) return rc;
    if (S == 0) return DLAGP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = wait_setup(ctx))) return rc;
    return sweep_from_host(ctx->self, S, kd, z, nhi, out);
} DLAGP_CATCH_INT("dlagp_eval")

int dlagp_null(dlagp_ctx *ctx, double *out) try {
    if (!ctx || !out) return fail(DLAGP_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = wait_setup(ctx)) return rc;
    return sweep_from_host(ctx->self, 1, 0, nullptr, nullptr, out);
} DLAGP_CATCH_INT("dlagp_null")

int dlagp_mvnpdf(int n, int k, const double *y, const double *mu, const double *M, ptrdiff_t m_row_stride,
                 ptrdiff_t m_col_stride, const double *d, double *out, int device) try {
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
} DLAGP_CATCH_INT("dlagp_mvnpdf")

int dlagp_absorption(dlagp_ctx *ctx, int kd, const double *z, const double *nhi, double *out) try {
    int rc = check_eval_args(ctx, 1, kd, z, nhi, out);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = wait_setup(ctx))) return rc;
    LaunchGroup &g = ctx->self;
    Workspace &ws = g.ws;
    if ((rc = grow(g, ws.z, kd * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.nhi, kd * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.out, (size_t)ctx->sl.n_pad * sizeof(double)))) return rc;
    HIP_TRY(hipMemcpy(ws.z.ptr, z, kd * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ws.nhi.ptr, nhi, kd * sizeof(double), hipMemcpyHostToDevice));
    const int lds_entries = std::max(ctx->sl.num_entries, ctx->sl.n_pad);
    const size_t lds_bytes = (size_t)(lds_entries + ctx->sl.n_pad + DLAGP_MAX_LINES + 1 + 8) * sizeof(double);
    if (lds_bytes > 160 * 1024) return fail(DLAGP_EINVAL, "sightline too long: %zu B of LDS needed", lds_bytes);
    if (lds_bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(absorption_generic),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(absorption_generic, dim3(1), dim3(DLAGP_GENERIC_THREADS), lds_bytes, nullptr,
                       ctx->self.d_table.as<SightlineDev>(), kd, ws.z.as<double>(), ws.nhi.as<double>(),
                       ws.out.as<double>(), lds_entries, ctx->sl.n_pad);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, ws.out.ptr, (size_t)ctx->sl.n * sizeof(double), hipMemcpyDeviceToHost));
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_absorption")

int dlagp_eval_device(dlagp_ctx *ctx, int64_t S, int kd, const double *d_z, const double *d_nhi, double *d_out,
                      void *stream) try {
    int rc = check_eval_args(ctx, S, kd, d_z, d_nhi, d_out);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = wait_setup(ctx))) return rc;
    return launch_sweep(ctx->self, S, kd, SampleSource{d_z, d_nhi, nullptr, 0, 0, 0, nullptr, nullptr, 0, 0, 0, 0, 0}, d_out,
                        static_cast<hipStream_t>(stream));
} DLAGP_CATCH_INT("dlagp_eval_device")

int dlagp_batch_create(dlagp_batch **out, dlagp_ctx *const *ctxs, int nb) try {
    if (!out) return fail(DLAGP_EINVAL, "out is NULL");
    *out = nullptr;
    if (!ctxs || nb <= 0) return fail(DLAGP_EINVAL, "empty batch");
    for (int b = 0; b < nb; ++b) {
        if (!ctxs[b]) return fail(DLAGP_EINVAL, "context %d is NULL", b);
        if (ctxs[b]->device != ctxs[0]->device)
            return fail(DLAGP_EINVAL, "context %d lives on device %d, batch on %d", b, ctxs[b]->device, ctxs[0]->device);
        if (ctxs[b]->sl.k != ctxs[0]->sl.k)
            return fail(DLAGP_EINVAL, "context %d has rank %d, batch rank is %d", b, ctxs[b]->user_k, ctxs[0]->user_k);
    }
    if (hipSetDevice(ctxs[0]->device) != hipSuccess) return fail(DLAGP_EDEVICE, "hipSetDevice(%d) failed", ctxs[0]->device);
    dlagp_batch *batch = new dlagp_batch();
    batch->group.device = ctxs[0]->device;
    batch->group.ctxs.assign(ctxs, ctxs + nb);
    {  // adopt the workspace the previous batch on this device left behind (see dlagp_batch_destroy)
        std::lock_guard<std::mutex> lock(g_spare_mutex);
        auto it = g_spare_workspace.find(batch->group.device);
        if (it != g_spare_workspace.end() && !it->second.empty()) {
            batch->group.ws = it->second.back();
            it->second.pop_back();
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_membership_mutex);
        for (dlagp_ctx *ctx : batch->group.ctxs) ctx->batches.push_back(batch);
    }
    // deferred members are complete once their own events have fired -- whichever thread created them
    int rc = DLAGP_OK;
    for (dlagp_ctx *ctx : batch->group.ctxs)
        if (!rc) rc = wait_setup(ctx);
    if (!rc) rc = finish_group(batch->group);  // synchronises the calling thread's set-up stream (the batch's table)
    if (!rc) rc = split_mixed_batch(batch->group);
    if (rc) {
        dlagp_batch_destroy(batch);
        return rc;
    }
    *out = batch;
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_batch_create")

// ---- f1, the data half: NullGP.set_data's normalisation, cuts, masks and padded grid (reference null_gp.py:117-177), the
// absorber-redshift bounds (set_parameters.py:125-159) and the sample tables (dla_samples.py:94-104), natively
static prep::Params prepare_params(const dlagp_prepare_params *p) {
    return prep::Params{p->min_lambda, p->max_lambda, p->normalization_min_lambda, p->normalization_max_lambda,
                        p->pixel_spacing, p->lya_wavelength, p->lyman_limit, p->min_z_cut, p->max_z_cut, p->width, p->normalize};
}

int dlagp_prepare_spectrum(const dlagp_prepare_params *params, int num_pixels, const double *wavelengths, const double *flux,
                           const double *noise_variance, const uint8_t *pixel_mask, double z_qso, const double *pad, int *n,
                           int *n_u, double *scalars, double *x, double *y, double *v, double *this_wavelengths,
                           double *unmasked_wavelengths, double *padded_wavelengths, uint8_t *keep) try {
    if (!params) return fail(DLAGP_EINVAL, "params is NULL");
    const prep::Params p = prepare_params(params);
    prep::Prepared out;
    const std::string err = prep::scan(p, prep::Spectrum{num_pixels, wavelengths, flux, noise_variance, pixel_mask, z_qso}, out);
    if (!err.empty()) return fail(DLAGP_EINVAL, "%s", err.c_str());
    std::vector<double> own(2 * (size_t)p.width);
    if (!pad) {
        prep::pad_libm(p, out.edge_lo, out.edge_hi, own.data());
        pad = own.data();
    }
    prep::place_padding(p, pad, out);
    if (n) *n = out.n;
    if (n_u) *n_u = out.n_u;
    if (scalars) {
        const double values[7] = {out.median, out.edge_lo, out.edge_hi, out.z_min, out.z_max, out.z_min_all, out.z_max_all};
        std::copy(values, values + 7, scalars);
    }
    auto give = [](double *dst, const std::vector<double> &src) { if (dst) std::copy(src.begin(), src.end(), dst); };
    give(x, out.x);
    give(y, out.y);
    give(v, out.v);
    give(this_wavelengths, out.this_wavelengths);
    give(unmasked_wavelengths, out.unmasked_wavelengths);
    give(padded_wavelengths, out.padded_wavelengths);
    if (keep) std::copy(out.keep.begin(), out.keep.end(), keep);
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_prepare_spectrum")

int dlagp_batch_prepare(dlagp_batch **out_batch, dlagp_ctx **out_ctxs, const dlagp_model *model,
                        const dlagp_prepare_params *params, int nb, const int *num_pixels, const double *const *wavelengths,
                        const double *const *flux, const double *const *noise_variance, const uint8_t *const *pixel_mask,
                        const double *z_qso, int num_lines, int flags, dlagp_pad_callback pad, void *pad_user, int64_t S,
                        const double *offsets, const double *sub_offsets, double *sample_z, double *sub_sample_z,
                        double *scalars) try {
    if (out_batch) *out_batch = nullptr;
    if (!out_batch || !out_ctxs || !model || !params || !num_pixels || !wavelengths || !flux || !noise_variance || !pixel_mask ||
        !z_qso)
        return fail(DLAGP_EINVAL, "NULL argument");
    if (nb <= 0) return fail(DLAGP_EINVAL, "empty batch");
    if (S < 0 || (S > 0 && ((sample_z && !offsets) || (sub_sample_z && !sub_offsets))))
        return fail(DLAGP_EINVAL, "a table of sample redshifts needs its offsets");
    if (flags & ~(DLAGP_FLAG_BROADENING | DLAGP_FLAG_LLS_BREAK)) return fail(DLAGP_EINVAL, "unknown flag bits 0x%x", flags);
    for (int b = 0; b < nb; ++b) out_ctxs[b] = nullptr;
    const prep::Params p = prepare_params(params);
    if (p.width != (DLAGP_NUM_TAPS - 1) / 2)
        return fail(DLAGP_EINVAL, "width = %d is not supported: the instrument profile has 2 * 3 + 1 taps", p.width);
    // 1. the index logic of every spectrum (host only)
    std::vector<prep::Prepared> prepared((size_t)nb);
    std::vector<double> edges(2 * (size_t)nb), padding(2 * (size_t)p.width * nb);
    for (int b = 0; b < nb; ++b) {
        const std::string err = prep::scan(
            p, prep::Spectrum{num_pixels[b], wavelengths[b], flux[b], noise_variance[b], pixel_mask[b], z_qso[b]}, prepared[b]);
        if (!err.empty()) return fail(DLAGP_EINVAL, "spectrum %d of the batch: %s", b, err.c_str());
        if (int rc = check_model_range(model, prepared[b].n, prepared[b].x.data())) return rc;
        if (prepared[b].n <= 0) return fail(DLAGP_EINVAL, "spectrum %d of the batch: every pixel of the modelled window is masked", b);
        edges[2 * b] = prepared[b].edge_lo;
        edges[2 * b + 1] = prepared[b].edge_hi;
    }
    // 2. the padding pixels: the caller's own log10 / power (NumPy's, for the whole batch at once), or libm's
    if (pad) pad(pad_user, nb, p.width, p.pixel_spacing, edges.data(), padding.data());
    else for (int b = 0; b < nb; ++b) prep::pad_libm(p, edges[2 * b], edges[2 * b + 1], padding.data() + 2 * (size_t)p.width * b);
    // 3. one context per spectrum (NullGP.get_interp on the device), none of them waited for, and the batch over them
    const bool broadening = (flags & DLAGP_FLAG_BROADENING) != 0;
    int rc = DLAGP_OK;
    for (int b = 0; b < nb && !rc; ++b) {
        prep::Prepared &sp = prepared[b];
        prep::place_padding(p, padding.data() + 2 * (size_t)p.width * b, sp);
        const double *grid = broadening ? sp.padded_wavelengths.data() : sp.unmasked_wavelengths.data();
        rc = create_context(&out_ctxs[b], sp.n, sp.n_u, model->dev.k, grid, sp.keep.data(), num_lines,
                            flags | DLAGP_FLAG_DEFER_SYNC, model->device, [&](dlagp_ctx *ctx) -> int {
            return fill_from_model(ctx, model, sp.n, sp.x.data(), sp.y.data(), sp.v.data(), sp.this_wavelengths.data(), z_qso[b]);
        });
    }
    if (!rc) rc = dlagp_batch_create(out_batch, out_ctxs, nb);
    if (rc) {
        const std::string message = g_last_error;  // the destructors below may overwrite it
        for (int b = 0; b < nb; ++b) {
            if (out_ctxs[b]) dlagp_destroy(out_ctxs[b]);
            out_ctxs[b] = nullptr;
        }
        g_last_error = message;
        return rc;
    }
    // 4. what the driver needs of the index logic
    for (int b = 0; b < nb; ++b) {
        const prep::Prepared &sp = prepared[b];
        if (sample_z && S) prep::sample_redshifts(sp.z_min, sp.z_max, offsets, S, sample_z + (size_t)b * S);
        if (sub_sample_z && S) prep::sample_redshifts(sp.z_min, sp.z_max, sub_offsets, S, sub_sample_z + (size_t)b * S);
        if (scalars) {
            const double values[8] = {sp.median, sp.z_min, sp.z_max, sp.z_min_all, sp.z_max_all, (double)sp.n, (double)sp.n_u, 0.0};
            std::copy(values, values + 8, scalars + 8 * (size_t)b);
        }
    }
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_batch_prepare")

void dlagp_batch_destroy(dlagp_batch *batch) try {
    if (!batch) return;
    (void)hipSetDevice(batch->group.device);
    // the workspace goes to the next batch, which may launch on another stream
    batch->group.drain();
    {
        std::lock_guard<std::mutex> lock(g_membership_mutex);
        for (dlagp_ctx *ctx : batch->group.ctxs)
            if (ctx) ctx->batches.erase(std::remove(ctx->batches.begin(), ctx->batches.end(), batch), ctx->batches.end());
    }
    // A catalogue run builds one batch after the other: the sweep workspace (partial sums, profile cache --
    // gigabytes) is handed to the next batch of this device instead of being freed and allocated again.
    // Up to `spare_workspaces` (4) per device are kept; option "pool_mb" = 0 turns this off as well.
    Workspace keep{};
    bool parked = false;
    if (g_opt_pool_mb > 0 && g_opt_spare_workspaces > 0) {
        std::lock_guard<std::mutex> lock(g_spare_mutex);
        std::vector<Workspace> &spares = g_spare_workspace[batch->group.device];
        if ((int64_t)spares.size() >= g_opt_spare_workspaces) {
            keep = spares.front();  // the oldest spare goes
            spares.erase(spares.begin());
        }
        spares.push_back(batch->group.ws);
        batch->group.ws = Workspace{};
        parked = true;
    }
    if (parked) keep.release();
    batch->group.release();
    delete batch;
} DLAGP_CATCH_VOID("dlagp_batch_destroy")

int dlagp_batch_eval_device(dlagp_batch *batch, int64_t S, int kd, const double *d_z, const double *d_nhi,
                            double *d_out, void *stream) try {
    int rc = check_eval_args(batch, S, kd, d_z, d_nhi, d_out);
    if (rc || (rc = batch_alive(batch))) return rc;
    cancel_pending(batch);
    HIP_TRY(hipSetDevice(batch->group.device));
    return launch_sweep(batch->group, S, kd, SampleSource{d_z, d_nhi, nullptr, 0, 0, 0, nullptr, nullptr, 0, 0, 0, 0, 0}, d_out,
                        static_cast<hipStream_t>(stream));
} DLAGP_CATCH_INT("dlagp_batch_eval_device")

// Log-likelihoods AND the level's evidence in one call (the by-sample pair dlagp_*eval_device + dlagp_level_evidence_device;
// reference dla_gp.py:132-190 for one occupation level): d_out receives log p - log S (NaN where too_close), d_evidence the
// log-mean-exp evidence.  Where the launch ends in finalize_rank20 the evidence is that kernel's tail (the last workgroup to
// arrive combines the blocks' sums: no second launch); plans that finalise in the accumulate waves, ranks above 20, mixed
// batches and the generic kernel get the level_evidence launch behind theirs.
static int eval_with_evidence(LaunchGroup &g, int64_t S, int kd, const double *d_z, const double *d_nhi, double *d_out, int level,
                              const uint8_t *d_too_close, double *d_evidence, hipStream_t stream) {
    if (!d_evidence) return fail(DLAGP_EINVAL, "d_evidence is NULL");
    if (level < 0) return fail(DLAGP_EINVAL, "level = %d, need level >= 0", level);
    TailRequest tail;
    tail.level = level;
    tail.too_close = d_too_close;
    tail.evidence = d_evidence;
    int rc = launch_sweep(g, S, kd, SampleSource{d_z, d_nhi, nullptr, 0, 0, 0, nullptr, nullptr, 0, 0, 0, 0, 0}, d_out, stream, &tail);
    if (rc) return rc;
    g_last_evidence_folded = tail.done ? 1 : 0;
    if (!tail.done) {
        hipLaunchKernelGGL(level_evidence, dim3((unsigned)g.ctxs.size()), dim3(DLAGP_EVIDENCE_THREADS), 0, stream, d_out, (long)S, level,
                           d_too_close, d_evidence, std::log((double)S));
        HIP_TRY(hipGetLastError());
    }
    return DLAGP_OK;
}

int dlagp_eval_evidence_device(dlagp_ctx *ctx, int64_t S, int kd, const double *d_z, const double *d_nhi, double *d_out,
                               int level, const uint8_t *d_too_close, double *d_evidence, void *stream) try {
    int rc = check_eval_args(ctx, S, kd, d_z, d_nhi, d_out);
    if (rc) return rc;
    if (S <= 0) return fail(DLAGP_EINVAL, "S = %lld, need S > 0", (long long)S);
    HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = wait_setup(ctx))) return rc;
    return eval_with_evidence(ctx->self, S, kd, d_z, d_nhi, d_out, level, d_too_close, d_evidence, static_cast<hipStream_t>(stream));
} DLAGP_CATCH_INT("dlagp_eval_evidence_device")

int dlagp_batch_eval_evidence_device(dlagp_batch *batch, int64_t S, int kd, const double *d_z, const double *d_nhi, double *d_out,
                                     int level, const uint8_t *d_too_close, double *d_evidence, void *stream) try {
    int rc = check_eval_args(batch, S, kd, d_z, d_nhi, d_out);
    if (rc || (rc = batch_alive(batch))) return rc;
    if (S <= 0) return fail(DLAGP_EINVAL, "S = %lld, need S > 0", (long long)S);
    cancel_pending(batch);
    HIP_TRY(hipSetDevice(batch->group.device));
    return eval_with_evidence(batch->group, S, kd, d_z, d_nhi, d_out, level, d_too_close, d_evidence, static_cast<hipStream_t>(stream));
} DLAGP_CATCH_INT("dlagp_batch_eval_evidence_device")

int dlagp_batch_eval(dlagp_batch *batch, int64_t S, int kd, const double *z, const double *nhi, double *out) try {
    int rc = check_eval_args(batch, S, kd, z, nhi, out);
    if (rc || (rc = batch_alive(batch))) return rc;
    cancel_pending(batch);
    if (S == 0) return DLAGP_OK;
    HIP_TRY(hipSetDevice(batch->group.device));
    return sweep_from_host(batch->group, S, kd, z, nhi, out);
} DLAGP_CATCH_INT("dlagp_batch_eval")

int dlagp_batch_null(dlagp_batch *batch, double *out) try {
    if (!batch || !out) return fail(DLAGP_EINVAL, "NULL argument");
    int rc = batch_alive(batch);
    if (rc) return rc;
    cancel_pending(batch);
    HIP_TRY(hipSetDevice(batch->group.device));
    return sweep_from_host(batch->group, 1, 0, nullptr, nullptr, out);
} DLAGP_CATCH_INT("dlagp_batch_null")

// the sweep about to be enqueued leaves its near-tie counts in the workspace (dlagp_batch_resample_ties fetches them)
static void ties_left_on_device(dlagp_batch *batch, int max_dlas, hipStream_t stream) {
    batch->ties_levels = max_dlas - 1;
    batch->ties_on_device = true;
    batch->ties_stream = stream;
}
// ... or brings them back with its results
static void ties_come_back(dlagp_batch *batch, Transfers &tr, int max_dlas) {
    const size_t count = batch->group.ctxs.size() * (size_t)(max_dlas - 1);
    batch->ties.assign(count, 0);
    batch->ties_levels = max_dlas - 1;
    batch->ties_on_device = false;
    if (count) tr.download(batch->ties.data(), batch->group.ws.ties.ptr, count * sizeof(int32_t));
}

static int check_sweep_args(const dlagp_batch *batch, int64_t S, int max_dlas, const void *sample_z,
                            const void *sample_nhi, int64_t nhi_stride, const void *uniforms, const void *evidences) {
    int alive = batch_alive(batch);
    if (alive) return alive;
    if (S <= 0) return fail(DLAGP_EINVAL, "S = %lld, need S > 0", (long long)S);
    if (S > INT32_MAX) return fail(DLAGP_EINVAL, "S = %lld does not fit the int32 base indices", (long long)S);
    if (max_dlas < 1 || max_dlas > DLAGP_MAX_ABSORBERS)
        return fail(DLAGP_EINVAL, "max_dlas = %d outside [1, %d]", max_dlas, DLAGP_MAX_ABSORBERS);
    if (nhi_stride != 0 && nhi_stride < S) return fail(DLAGP_EINVAL, "nhi_stride must be 0 (shared) or >= S");
    if (!sample_z || !sample_nhi || !evidences) return fail(DLAGP_EINVAL, "NULL argument");
    if (max_dlas > 1 && !uniforms) return fail(DLAGP_EINVAL, "uniforms are required when max_dlas > 1");
    return DLAGP_OK;
}

int dlagp_batch_sweep_device(dlagp_batch *batch, int64_t S, int max_dlas, const double *d_sample_z,
                             const double *d_sample_nhi, int64_t nhi_stride, double min_z_separation,
                             const double *d_uniforms, double *d_evidences, double *d_sample_ll, int32_t *d_base_inds,
                             void *stream) try {
    CallOptionsScope options_of_this_call;  // one reading of the launch-shaping options for the whole call
    int rc = check_sweep_args(batch, S, max_dlas, d_sample_z, d_sample_nhi, nhi_stride, d_uniforms, d_evidences);
    if (rc) return rc;
    if (!d_sample_ll || (max_dlas > 1 && !d_base_inds))
        return fail(DLAGP_EINVAL, "the device variant needs d_sample_ll and d_base_inds (they are its working state)");
    cancel_pending(batch);
    HIP_TRY(hipSetDevice(batch->group.device));
    ties_left_on_device(batch, max_dlas, static_cast<hipStream_t>(stream));
    return sweep_levels(batch->group, S, max_dlas, d_sample_z, d_sample_nhi, nhi_stride, min_z_separation, d_uniforms,
                        d_evidences, d_sample_ll, d_base_inds, static_cast<hipStream_t>(stream));
} DLAGP_CATCH_INT("dlagp_batch_sweep_device")

int dlagp_batch_two_model_sweep_device(dlagp_batch *batch, int64_t S, int max_dlas, const double *d_sample_z,
                                       const double *d_sub_nhi, int64_t sub_nhi_stride, const double *d_sample_nhi,
                                       int64_t nhi_stride, double min_z_separation, const double *d_uniforms,
                                       double *d_sub_evidences, double *d_sub_sample_ll, double *d_evidences,
                                       double *d_sample_ll, int32_t *d_base_inds, void *stream_) try {
    CallOptionsScope options_of_this_call;  // one reading of the launch-shaping options for the whole call
    int rc = check_sweep_args(batch, S, max_dlas, d_sample_z, d_sample_nhi, nhi_stride, d_uniforms, d_evidences);
    if (rc) return rc;
    if ((rc = check_sweep_args(batch, S, 1, d_sample_z, d_sub_nhi, sub_nhi_stride, nullptr, d_sub_evidences))) return rc;
    if (!d_sample_ll || !d_sub_sample_ll || (max_dlas > 1 && !d_base_inds))
        return fail(DLAGP_EINVAL, "the device variant needs the sample_ll arrays and d_base_inds (they are its working state)");
    cancel_pending(batch);
    LaunchGroup &g = batch->group;
    HIP_TRY(hipSetDevice(g.device));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if ((rc = enter_stream(g, stream))) return rc;
    ties_left_on_device(batch, max_dlas, stream);
    if (pair_eligible(g, S)) {
        const PairSpec sub{d_sub_nhi, sub_nhi_stride, d_sub_evidences, d_sub_sample_ll};
        return sweep_levels(g, S, max_dlas, d_sample_z, d_sample_nhi, nhi_stride, min_z_separation, d_uniforms, d_evidences,
                            d_sample_ll, d_base_inds, stream, false, &sub);
    }
    // not a pair launch's shape (a single sightline, a mixed batch, another line count ...): the two sweeps one after the other
    rc = sweep_levels(g, S, 1, d_sample_z, d_sub_nhi, sub_nhi_stride, min_z_separation, nullptr, d_sub_evidences, d_sub_sample_ll,
                      nullptr, stream);
    if (rc) return rc;
    return sweep_levels(g, S, max_dlas, d_sample_z, d_sample_nhi, nhi_stride, min_z_separation, d_uniforms, d_evidences,
                        d_sample_ll, d_base_inds, stream, /*keep_sticky_flag=*/true);
} DLAGP_CATCH_INT("dlagp_batch_two_model_sweep_device")

int dlagp_batch_sweep(dlagp_batch *batch, int64_t S, int max_dlas, const double *sample_z, const double *sample_nhi,
                      int64_t nhi_stride, double min_z_separation, const double *uniforms, double *evidences,
                      double *sample_ll, int32_t *base_inds) try {
    CallOptionsScope options_of_this_call;  // one reading of the launch-shaping options for the whole call
    int rc = check_sweep_args(batch, S, max_dlas, sample_z, sample_nhi, nhi_stride, uniforms, evidences);
    if (rc) return rc;
    cancel_pending(batch);
    LaunchGroup &g = batch->group;
    HIP_TRY(hipSetDevice(g.device));
    Workspace &ws = g.ws;
    const size_t nb = g.ctxs.size();
    const size_t rows = nb * (size_t)S;
    const size_t n_nhi = nhi_stride ? (size_t)nhi_stride * (nb - 1) + S : (size_t)S;
    const size_t n_base = rows * (max_dlas - 1);
    if ((rc = grow(g, ws.z, rows * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.nhi, n_nhi * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.uniforms, std::max<size_t>(n_base, 1) * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.evidences, nb * max_dlas * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.sample_ll, rows * max_dlas * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.base, std::max<size_t>(n_base, 1) * sizeof(int32_t)))) return rc;
    Transfers tr(ws, g.device);
    const hipStream_t stream = tr.compute;
    tr.upload(ws.z.ptr, sample_z, rows * sizeof(double));
    tr.upload(ws.nhi.ptr, sample_nhi, n_nhi * sizeof(double));
    tr.upload(ws.uniforms.ptr, uniforms, n_base * sizeof(double));
    tr.download(evidences, ws.evidences.ptr, nb * max_dlas * sizeof(double));
    tr.download(sample_ll, ws.sample_ll.ptr, rows * max_dlas * sizeof(double));
    tr.download(base_inds, ws.base.ptr, n_base * sizeof(int32_t));
    int status = 0;
    if ((rc = grow(g, ws.alive, (nb + 1) * sizeof(int)))) return rc;  // [nb]: the sweep's sticky non-PD flag (sweep_levels)
    if ((rc = grow(g, ws.ties, nb * std::max(max_dlas - 1, 1) * sizeof(int)))) return rc;
    tr.download(&status, ws.alive.as<int>() + nb, sizeof(int));
    ties_come_back(batch, tr, max_dlas);
    if ((rc = tr.reserve()) || (rc = tr.run_uploads())) return rc;
    rc = sweep_levels(g, S, max_dlas, ws.z.as<double>(), ws.nhi.as<double>(), nhi_stride, min_z_separation,
                      ws.uniforms.as<double>(), ws.evidences.as<double>(), ws.sample_ll.as<double>(),
                      ws.base.as<int>(), stream);
    if (rc) return rc;
    if ((rc = tr.run_downloads())) return rc;
    g.launched = false;  // as in sweep_from_host
    for (auto &p : g.parts) p->launched = false;
    if (status) return fail(DLAGP_ELINALG, "Matrix is not positive definite (I + M' D^-1 M)");
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_batch_sweep")

// The evidence loop of BayesModelSelect.model_selection (reference bayesian_model_selection.py:84-95) for every
// sightline of the batch in ONE call: the one-absorber model (sub-DLAs), the occupation sweep of the DLA model and the
// null model are enqueued back to back and waited for once -- one upload phase, one wait and one download phase per
// batch instead of three of each.
int dlagp_batch_model_evidences_start(dlagp_batch *batch, int64_t S, int max_dlas, const double *sub_sample_z,
                                      const double *sub_sample_nhi, int64_t sub_nhi_stride, const double *sample_z,
                                      const double *sample_nhi, int64_t nhi_stride, double min_z_separation) try {
    CallOptionsScope options_of_this_call;  // one reading of the launch-shaping options for the whole call
    double dummy = 0.0;  // check_sweep_args wants the output arrays of the one-call form: they come with _finish
    int rc = check_sweep_args(batch, S, max_dlas, sample_z, sample_nhi, nhi_stride, max_dlas > 1 ? &dummy : nullptr, &dummy);
    if (rc) return rc;
    if ((rc = check_sweep_args(batch, S, 1, sub_sample_z, sub_sample_nhi, sub_nhi_stride, nullptr, &dummy))) return rc;
    LaunchGroup &g = batch->group;
    HIP_TRY(hipSetDevice(g.device));
    Workspace &ws = g.ws;
    cancel_pending(batch);  // a call that was started and never finished: let its kernels drain
    const size_t nb = g.ctxs.size();
    const size_t rows = nb * (size_t)S;
    const size_t n_nhi = nhi_stride ? (size_t)nhi_stride * (nb - 1) + S : (size_t)S;
    const size_t n_sub_nhi = sub_nhi_stride ? (size_t)sub_nhi_stride * (nb - 1) + S : (size_t)S;
    const size_t n_base = rows * (max_dlas - 1);
    if ((rc = grow(g, ws.z, rows * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.nhi, n_nhi * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.uniforms, std::max<size_t>(n_base, 1) * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.evidences, nb * max_dlas * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.sample_ll, rows * max_dlas * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.base, std::max<size_t>(n_base, 1) * sizeof(int32_t)))) return rc;
    if ((rc = grow(g, ws.sub_z, rows * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.sub_nhi, n_sub_nhi * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.sub_evidences, nb * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.sub_ll, rows * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.out, nb * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.alive, (nb + 1) * sizeof(int)))) return rc;  // [nb]: the sweeps' sticky non-PD flag
    if ((rc = grow(g, ws.ties, nb * std::max(max_dlas - 1, 1) * sizeof(int)))) return rc;
    if ((rc = grow(g, ws.status, sizeof(int)))) return rc;            // the null launch's
    if ((rc = grow(g, ws.sub_level_ll, rows * sizeof(double)))) return rc;
    if ((rc = grow(g, ws.sub_alive, (nb + 1) * sizeof(int)))) return rc;
    // Both models at the same redshift samples (the reference's loaders build them by one formula from one set of offsets,
    // dla_samples.py:94-104 / subdla_samples.py:115-125; compared by content here): the sub-DLA sweep rides in level 1's
    // launch of the DLA sweep, one optical depth per QMC point for both (pair launch).
    const bool pair = std::memcmp(sub_sample_z, sample_z, rows * sizeof(double)) == 0 && pair_eligible(g, S);
    Transfers tr(ws, g.device);
    const hipStream_t stream = tr.compute;
    if (!pair) tr.upload(ws.sub_z.ptr, sub_sample_z, rows * sizeof(double));
    tr.upload(ws.sub_nhi.ptr, sub_sample_nhi, n_sub_nhi * sizeof(double));
    tr.upload(ws.z.ptr, sample_z, rows * sizeof(double));
    tr.upload(ws.nhi.ptr, sample_nhi, n_nhi * sizeof(double));
    // the work items of the pair launch, planned now: they go up with the samples instead of as a blocking copy of their own
    // between the call's launches (15 us of a one-spectrum call, all of it in front of the first sweep kernel)
    std::vector<char> plan_bytes;
    if (pair) {
        LaunchPlan *plan = nullptr;
        if ((rc = build_plan(g, pair_virtual_samples(S), &plan, &plan_bytes))) return rc;
        if (!plan_bytes.empty()) tr.upload(plan->blob.ptr, plan_bytes.data(), plan_bytes.size());
    }
    // the staging buffer is sized for BOTH phases now: growing it in _finish would free memory the copies of this
    // phase may still be reading
    const size_t second_phase = (n_base + 2 * nb + rows + nb * max_dlas + rows * max_dlas) * sizeof(double) +
                                n_base * sizeof(int32_t) + 2 * sizeof(int) + nb * max_dlas * sizeof(int32_t) + 16 * 256;
    if ((rc = tr.reserve(second_phase)) || (rc = tr.run_uploads())) {
        if (!plan_bytes.empty()) {  // the new plan's items never reached the device: no later launch may find it
            g.drain();  // (launches in flight still read the other plans)
            (void)hipStreamSynchronize(tr.transfer);  // ... and whatever part of the uploads did leave may be writing this one
            for (auto &kv : g.plans) kv.second->release();
            g.plans.clear();
        }
        return rc;
    }
    if (!plan_bytes.empty()) {
        g.upload_stream = tr.transfer;
        g.upload_pending = true;
    }
    // The null model of every member BESIDE the sweeps: on this thread's transfer stream, with buffers of its own -- 30 us of a
    // one-spectrum call that nothing has to wait for.  When?  Not in front of the sweep's first launch: on a one-spectrum call
    // everything there is the host's time with the device idle (two launches and their events: ~20 us).  Not right behind that
    // launch either: the generic kernel's workgroups take compute units from an accumulate launch whose work items are dealt
    // to the wave slots in advance, and the launch waits for its slowest unit (+22 us on a 237 us pair launch; the null
    // kernel itself took 267 us instead of 24).  _finish enqueues it behind an event that fires when the deepest level's
    // evaluations are done: it runs beside that level's finalisation and level finish, small kernels that leave most of the
    // chip idle.  (A one-level call has no such moment: there it goes first, as it did before.)
    // Only for a few sightlines (a caller that works one spectrum at a time): beside OTHER batches' sweeps, as in the
    // catalogue pipeline, the generic kernel's four-wave workgroups would wait for four free wave slots on one compute unit
    // (host_shared.h, DLAGP_SETUP_THREADS) and hold this batch's downloads back; there the null model stays the call's
    // last launch on the kernel stream (45 us of a 12 ms batch).
    batch->pending.null_on_side = g_opt_null_side != 0 && nb <= 4;
    batch->pending.null_launched = false;
    // its flag exists from here on: _finish registers the flag's download before the launch that would allocate it
    if (batch->pending.null_on_side && (rc = ws.null_status.reserve(sizeof(int)))) return rc;
    if (batch->pending.null_on_side && max_dlas == 1) {
        if ((rc = launch_null_side(g, ws.out.as<double>(), tr.transfer))) return rc;
        batch->pending.null_launched = true;
    }
    batch->pending.pair = pair;
    batch->pending.sub_nhi_stride = sub_nhi_stride;
    if (pair) {
        const PairSpec sub{ws.sub_nhi.as<double>(), sub_nhi_stride, ws.sub_evidences.as<double>(), ws.sub_ll.as<double>()};
        rc = sweep_levels(g, S, max_dlas, ws.z.as<double>(), ws.nhi.as<double>(), nhi_stride, min_z_separation, nullptr,
                          ws.evidences.as<double>(), ws.sample_ll.as<double>(), ws.base.as<int>(), stream, false, &sub,
                          SWEEP_START, &batch->pending.cached);
    } else {
        // model order of the reference's model list (run_bayes_select.py:189): the one-absorber model before the DLA model
        rc = sweep_levels(g, S, 1, ws.sub_z.as<double>(), ws.sub_nhi.as<double>(), sub_nhi_stride, min_z_separation, nullptr,
                          ws.sub_evidences.as<double>(), ws.sub_ll.as<double>(), nullptr, stream);
    }
    if (rc) return rc;
    if (rc) return rc;
    batch->pending.active = true;
    batch->pending.S = S;
    batch->pending.max_dlas = max_dlas;
    batch->pending.nhi_stride = nhi_stride;
    batch->pending.min_z_separation = min_z_separation;
    batch->pending.stage_used = tr.total;
    batch->pending.stream = stream;
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_batch_model_evidences_start")

int dlagp_batch_model_evidences_finish(dlagp_batch *batch, const double *uniforms, double *null_evidences,
                                       double *sub_evidences, double *sub_sample_ll, double *evidences, double *sample_ll,
                                       int32_t *base_inds) try {
    CallOptionsScope options_of_this_call;  // one reading of the launch-shaping options for the whole call
    int rc = batch_alive(batch);
    if (rc) return rc;
    if (!batch->pending.active) return fail(DLAGP_EINVAL, "dlagp_batch_model_evidences_finish without a matching _start");
    const int64_t S = batch->pending.S;
    const int max_dlas = batch->pending.max_dlas;
    if (!null_evidences || !sub_evidences || !evidences || (max_dlas > 1 && !uniforms)) return fail(DLAGP_EINVAL, "NULL argument");
    LaunchGroup &g = batch->group;
    HIP_TRY(hipSetDevice(g.device));
    Workspace &ws = g.ws;
    const size_t nb = g.ctxs.size();
    const size_t rows = nb * (size_t)S;
    const size_t n_base = rows * (max_dlas - 1);
    Transfers tr(ws, g.device);
    const hipStream_t stream = tr.compute;
    if (stream != batch->pending.stream)
        return fail(DLAGP_EINVAL, "dlagp_batch_model_evidences_finish must be called by the thread that called _start");
    batch->pending.active = false;
    tr.base = batch->pending.stage_used;
    tr.upload(ws.uniforms.ptr, uniforms, n_base * sizeof(double));
    tr.download(null_evidences, ws.out.ptr, nb * sizeof(double));
    tr.download(sub_evidences, ws.sub_evidences.ptr, nb * sizeof(double));
    tr.download(sub_sample_ll, ws.sub_ll.ptr, rows * sizeof(double));
    tr.download(evidences, ws.evidences.ptr, nb * max_dlas * sizeof(double));
    tr.download(sample_ll, ws.sample_ll.ptr, rows * max_dlas * sizeof(double));
    tr.download(base_inds, ws.base.ptr, n_base * sizeof(int32_t));
    int status[2] = {0, 0};
    tr.download(&status[0], ws.alive.as<int>() + nb, sizeof(int));
    tr.download(&status[1], batch->pending.null_on_side ? ws.null_status.ptr : ws.status.ptr, sizeof(int));
    ties_come_back(batch, tr, max_dlas);
    const size_t capacity = ws.stage.bytes;
    if ((rc = tr.reserve())) return rc;
    if (ws.stage.bytes != capacity) return fail(DLAGP_EDEVICE, "internal: the staging buffer grew between the two phases");
    if ((rc = tr.run_uploads())) return rc;
    // the null model beside the deepest level's finalisation and level finish (see _start)
    const std::function<int()> null_beside_last_level = [&]() -> int {
        if (tr.transfer != stream) {
            CallEvents *ev = nullptr;
            int rc_ev = call_events(g.device, &ev);
            if (rc_ev) return rc_ev;
            if (!ev->gate) HIP_TRY(hipEventCreateWithFlags(&ev->gate, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ev->gate, stream));
            HIP_TRY(hipStreamWaitEvent(tr.transfer, ev->gate, 0));
        }
        batch->pending.null_launched = true;
        return launch_null_side(g, ws.out.as<double>(), tr.transfer);
    };
    const std::function<int()> *hook =
        batch->pending.null_on_side && !batch->pending.null_launched ? &null_beside_last_level : nullptr;
    if (batch->pending.pair) {
        const PairSpec sub{ws.sub_nhi.as<double>(), batch->pending.sub_nhi_stride, ws.sub_evidences.as<double>(), ws.sub_ll.as<double>()};
        rc = sweep_levels(g, S, max_dlas, ws.z.as<double>(), ws.nhi.as<double>(), batch->pending.nhi_stride,
                          batch->pending.min_z_separation, ws.uniforms.as<double>(), ws.evidences.as<double>(),
                          ws.sample_ll.as<double>(), ws.base.as<int>(), stream, /*keep_sticky_flag=*/true, &sub, SWEEP_FINISH,
                          &batch->pending.cached, hook);
    } else {
        rc = sweep_levels(g, S, max_dlas, ws.z.as<double>(), ws.nhi.as<double>(), batch->pending.nhi_stride,
                          batch->pending.min_z_separation, ws.uniforms.as<double>(), ws.evidences.as<double>(),
                          ws.sample_ll.as<double>(), ws.base.as<int>(), stream, /*keep_sticky_flag=*/true, nullptr, SWEEP_ALL,
                          nullptr, hook);
    }
    if (rc) return rc;
    if (batch->pending.null_on_side && !batch->pending.null_launched) {  // no level was launched by this phase
        if ((rc = null_beside_last_level())) return rc;
    }
    if (!batch->pending.null_on_side) {
        // the null model last: its launch leaves its own flag in ws.status, which the sweeps' launches would reset
        rc = launch_sweep(g, 1, 0, SampleSource{nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, 0, 0, 0, 0, 0},
                          ws.out.as<double>(), stream);
        if (rc) return rc;
    }  // (else: enqueued above on the transfer stream, which the downloads below follow on)
    if ((rc = tr.run_downloads())) return rc;
    g.launched = false;  // as in sweep_from_host
    g.side_launched = false;  // run_downloads waited for the transfer stream
    g.upload_pending = false;
    for (auto &p : g.parts) p->launched = false;
    if (status[0] || status[1]) return fail(DLAGP_ELINALG, "Matrix is not positive definite (I + M' D^-1 M)");
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_batch_model_evidences_finish")

int dlagp_batch_model_evidences_cancel(dlagp_batch *batch) try {
    if (!batch) return fail(DLAGP_EINVAL, "NULL batch");
    (void)hipSetDevice(batch->group.device);
    cancel_pending(batch);
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_batch_model_evidences_cancel")

int dlagp_batch_model_evidences(dlagp_batch *batch, int64_t S, int max_dlas, const double *sub_sample_z,
                                const double *sub_sample_nhi, int64_t sub_nhi_stride, const double *sample_z,
                                const double *sample_nhi, int64_t nhi_stride, double min_z_separation,
                                const double *uniforms, double *null_evidences, double *sub_evidences,
                                double *sub_sample_ll, double *evidences, double *sample_ll, int32_t *base_inds) try {
    if (max_dlas > 1 && !uniforms) return fail(DLAGP_EINVAL, "uniforms are required when max_dlas > 1");
    if (!null_evidences || !sub_evidences || !evidences) return fail(DLAGP_EINVAL, "NULL argument");
    int rc = dlagp_batch_model_evidences_start(batch, S, max_dlas, sub_sample_z, sub_sample_nhi, sub_nhi_stride, sample_z,
                                               sample_nhi, nhi_stride, min_z_separation);
    if (rc) return rc;
    return dlagp_batch_model_evidences_finish(batch, uniforms, null_evidences, sub_evidences, sub_sample_ll, evidences,
                                              sample_ll, base_inds);
} DLAGP_CATCH_INT("dlagp_batch_model_evidences")

int dlagp_batch_resample_ties(dlagp_batch *batch, int32_t *ties, int64_t capacity) try {
    int rc = batch_alive(batch);
    if (rc) return rc;
    if (batch->ties_levels < 0) return fail(DLAGP_EINVAL, "no occupation sweep has run on this batch");
    const size_t count = batch->group.ctxs.size() * (size_t)batch->ties_levels;
    if (count && (!ties || capacity < (int64_t)count))
        return fail(DLAGP_EINVAL, "ties needs room for %lld counts, has %lld", (long long)count, (long long)capacity);
    if (!count) return DLAGP_OK;
    if (batch->ties_on_device) {  // a *_device sweep: its kernels first
        HIP_TRY(hipSetDevice(batch->group.device));
        HIP_TRY(hipStreamSynchronize(batch->ties_stream));
        batch->ties.assign(count, 0);
        HIP_TRY(hipMemcpy(batch->ties.data(), batch->group.ws.ties.ptr, count * sizeof(int32_t), hipMemcpyDeviceToHost));
        batch->ties_on_device = false;
    }
    std::memcpy(ties, batch->ties.data(), count * sizeof(int32_t));
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_batch_resample_ties")

// ---- one sightline's samples split over several GPUs (SURVEY 8e "secondary", reference dla_gp.py:132-159 is
// independent per sample, :209-218 needs all S): every rank evaluates its slice of a level, the slices are
// all-gathered by the caller (RCCL), and every rank finishes the level on the full vector with the same uniforms
int dlagp_batch_level_eval_device(dlagp_batch *batch, int64_t S_total, int64_t s_begin, int64_t s_count, int level,
                                  const double *d_sample_z, const double *d_sample_nhi, int64_t nhi_stride,
                                  const int32_t *d_base_inds, int base_levels, double *d_out, void *stream) try {
    int alive = batch_alive(batch);
    if (alive) return alive;
    if (S_total <= 0 || s_begin < 0 || s_count < 0 || s_begin + s_count > S_total)
        return fail(DLAGP_EINVAL, "sample slice [%lld, %lld) outside [0, %lld)", (long long)s_begin,
                    (long long)(s_begin + s_count), (long long)S_total);
    if (level < 0 || level >= DLAGP_MAX_ABSORBERS) return fail(DLAGP_EINVAL, "level %d outside [0, %d)", level, DLAGP_MAX_ABSORBERS);
    if (level > base_levels) return fail(DLAGP_EINVAL, "level %d needs %d rows of base indices, have %d", level, level, base_levels);
    if (nhi_stride != 0 && nhi_stride < S_total) return fail(DLAGP_EINVAL, "nhi_stride must be 0 (shared) or >= S_total");
    if (!d_sample_z || !d_sample_nhi || (level > 0 && !d_base_inds) || (s_count > 0 && !d_out))
        return fail(DLAGP_EINVAL, "NULL argument");
    if (s_count == 0) return DLAGP_OK;
    HIP_TRY(hipSetDevice(batch->group.device));
    cancel_pending(batch);
    const SampleSource src{d_sample_z, d_sample_nhi, d_base_inds, 1, (long)nhi_stride, base_levels, nullptr, nullptr,
                           0, 0, 0, (long)s_begin, (long)S_total};
    return launch_sweep(batch->group, s_count, level + 1, src, d_out, static_cast<hipStream_t>(stream));
} DLAGP_CATCH_INT("dlagp_batch_level_eval_device")

int dlagp_sweep_state_init_device(int nb, int64_t S, int max_dlas, double *d_sample_ll, double *d_evidences,
                                  int32_t *d_base_inds, int32_t *d_flags, int device, void *stream_) try {
    if (nb <= 0 || S <= 0 || max_dlas < 1 || max_dlas > DLAGP_MAX_ABSORBERS) return fail(DLAGP_EINVAL, "bad shape");
    if (!d_sample_ll || !d_evidences || !d_flags || (max_dlas > 1 && !d_base_inds)) return fail(DLAGP_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(device));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const double nan = std::numeric_limits<double>::quiet_NaN();
    const long n_ll = (long)nb * S * max_dlas, n_base = (long)nb * S * (max_dlas - 1);
    hipLaunchKernelGGL(fill_f64, dim3((unsigned)std::min<long>((n_ll + 255) / 256, 2048)), dim3(256), 0, stream,
                       d_sample_ll, n_ll, nan);
    hipLaunchKernelGGL(fill_f64, dim3(1), dim3(256), 0, stream, d_evidences, (long)nb * max_dlas, nan);
    hipLaunchKernelGGL(fill_i32, dim3(1), dim3(256), 0, stream, d_flags, (long)nb, 1);
    hipLaunchKernelGGL(fill_i32, dim3(1), dim3(64), 0, stream, d_flags + nb, 2L + (long)nb * (max_dlas - 1), 0);
    if (n_base > 0)
        hipLaunchKernelGGL(fill_i32, dim3((unsigned)std::min<long>((n_base + 255) / 256, 2048)), dim3(256), 0, stream,
                           d_base_inds, n_base, 0);
    HIP_TRY(hipGetLastError());
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_sweep_state_init_device")

int dlagp_level_finish_device(int nb, int64_t S, int level, int max_dlas, const double *d_ll, const double *d_sample_z,
                              double min_z_separation, const double *d_uniforms, double *d_sample_ll,
                              double *d_evidences, int32_t *d_base_inds, double *d_cdf_scratch, int32_t *d_flags,
                              int device, void *stream) try {
    if (nb <= 0 || S <= 0 || level < 0 || level >= max_dlas || max_dlas > DLAGP_MAX_ABSORBERS)
        return fail(DLAGP_EINVAL, "bad shape");
    if (!d_ll || !d_sample_z || !d_sample_ll || !d_evidences || !d_cdf_scratch || !d_flags ||
        (max_dlas > 1 && (!d_base_inds || !d_uniforms)))
        return fail(DLAGP_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(device));
    // the kernel reads the redshifts of a sample's absorbers through the index table; column densities are not used
    const SampleSource src{d_sample_z, d_sample_z, d_base_inds, 1, 0, max_dlas - 1, nullptr, nullptr, 0, 0, 0, 0, 0};
    static_assert(DLAGP_LEVEL_EXCHANGE == DLAGP_LEVEL_EXCH, "include/dlagp.h and dlagp_device.h disagree on the scratch size");
    const LevelArgs args{d_ll, (long)S, level, max_dlas, src, min_z_separation, std::log((double)S), d_uniforms, d_sample_ll,
                         d_evidences, d_base_inds, d_cdf_scratch, d_cdf_scratch + (size_t)nb * S, d_flags, d_flags + nb + 1,
                         d_flags + nb, max_dlas > 1 ? d_flags + nb + 2 : nullptr};
    return launch_level_finish(nb, static_cast<hipStream_t>(stream), args);
} DLAGP_CATCH_INT("dlagp_level_finish_device")

int dlagp_level_evidence_device(double *d_ll, int nb, int64_t S, int level, const uint8_t *d_too_close,
                                double *d_evidence, int device, void *stream) try {
    if (!d_ll || !d_evidence) return fail(DLAGP_EINVAL, "NULL argument");
    if (nb <= 0 || S <= 0 || level < 0) return fail(DLAGP_EINVAL, "need nb > 0, S > 0, level >= 0");
    HIP_TRY(hipSetDevice(device));
    hipLaunchKernelGGL(level_evidence, dim3((unsigned)nb), dim3(DLAGP_EVIDENCE_THREADS), 0,
                       static_cast<hipStream_t>(stream), d_ll, (long)S, level, d_too_close, d_evidence,
                       std::log((double)S));
    HIP_TRY(hipGetLastError());
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_level_evidence_device")

int dlagp_debug_voigt(int form, int line, int64_t n, const double *velocity, double *out, int device) try {
    if (!velocity || !out || n < 0) return fail(DLAGP_EINVAL, "bad argument");
    if (form < 0 || form > 1) return fail(DLAGP_EINVAL, "form must be 0 (generic kernels) or 1 (MFMA kernel)");
    if (line < 0 || line >= DLAGP_MAX_LINES) return fail(DLAGP_EINVAL, "line %d outside [0, %d)", line, DLAGP_MAX_LINES);
    if (n == 0) return DLAGP_OK;
    int rc = ensure_device(device);
    if (rc) return rc;
    DeviceBuffer d_in, d_out;
    if ((rc = d_in.reserve((size_t)n * sizeof(double))) || (rc = d_out.reserve((size_t)n * sizeof(double)))) {
        d_in.release();
        return rc;
    }
    hipError_t err = hipMemcpy(d_in.ptr, velocity, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        hipLaunchKernelGGL(voigt_forms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, form, line, (long)n,
                           d_in.as<const double>(), 1.0 / (std::sqrt(2.0 * M_PI) * DLAGP_SIGMA_CGS), d_out.as<double>());
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpy(out, d_out.ptr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    d_in.release();
    d_out.release();
    if (err != hipSuccess) return fail(DLAGP_EDEVICE, "dlagp_debug_voigt failed: %s", hipGetErrorString(err));
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_debug_voigt")

int dlagp_debug_read(uint64_t *out, int n, int reset) try {
#ifdef DLAGP_TRACE_BUILD
    if (n > 8) {  // diagnostic build: the per-item trace of the last launches (kernels_mfma.h)
        std::vector<unsigned long long> trace(4 * 8192);
        HIP_TRY(hipMemcpyFromSymbol(trace.data(), HIP_SYMBOL(DLAGP_TRACE), trace.size() * sizeof(unsigned long long)));
        for (int i = 0; i < n && i < (int)trace.size(); ++i) out[i] = trace[i];
        return DLAGP_OK;
    }
#endif
    unsigned long long host[8] = {0};
    HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(DLAGP_STAMPS), sizeof(host)));
    for (int i = 0; i < n && i < 8; ++i)
        if (out) out[i] = host[i];
    if (reset) {
        unsigned long long zero[8] = {0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(DLAGP_STAMPS), zero, sizeof(zero)));
    }
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_debug_read")

int dlagp_set_option(const char *name, int64_t value) try {
    if (!name) return fail(DLAGP_EINVAL, "option name is NULL");
    if (!strcmp(name, "pixel_splits")) g_opt_pixel_splits = value;
    else if (!strcmp(name, "path")) g_opt_path = value;
    else if (!strcmp(name, "timing")) g_opt_timing = value;
    else if (!strcmp(name, "debug")) g_opt_debug = value;
    else if (!strcmp(name, "core_patch")) g_opt_core_patch = value;
    else if (!strcmp(name, "pool_mb")) {
        g_opt_pool_mb = value;
        if (value <= 0) {
            {
                // the pack buffers of the set-up lanes no thread holds (their streams drained first)
                LanePool &lanes = lane_pool();
                std::lock_guard<std::mutex> lock(lanes.mutex);
                for (auto &kv : lanes.free) {
                    (void)hipSetDevice(kv.first);
                    for (SetupLane *lane : kv.second) {
                        (void)hipStreamSynchronize(lane->stream);
                        lane->pack.release();
                    }
                }
            }
            pool_flush();
            std::lock_guard<std::mutex> lock(g_spare_mutex);
            for (auto &kv : g_spare_workspace) {
                (void)hipSetDevice(kv.first);
                for (Workspace &w : kv.second) w.release();
            }
            g_spare_workspace.clear();
        }
    }
    else if (!strcmp(name, "profile_cache")) g_opt_profile_cache = value;
    else if (!strcmp(name, "profile_cache_max_mb")) g_opt_profile_cache_max_mb = value;
    else if (!strcmp(name, "plan_item_cost")) g_opt_plan_item_cost = value;
    else if (!strcmp(name, "plan_rest_first")) g_opt_plan_rest_first = value;
    else if (!strcmp(name, "plan_slots")) g_opt_plan_slots = value > 0 ? value : 2048;
    else if (!strcmp(name, "plan_xcd_affine")) g_opt_plan_xcd_affine = value;
    else if (!strcmp(name, "plan_taper")) g_opt_plan_taper = value < 1 ? 1 : (value > 8 ? 8 : value);
    else if (!strcmp(name, "plan_affine_min_chunks")) g_opt_plan_affine_min_chunks = value < 0 ? 0 : value;
    else if (!strcmp(name, "finalize_row16_blocks")) g_opt_finalize_row16_blocks = value;
    else if (!strcmp(name, "spare_workspaces")) g_opt_spare_workspaces = value;
    else if (!strcmp(name, "shared_work_stream")) g_opt_shared_work_stream = value != 0;
    else if (!strcmp(name, "shared_rows")) g_opt_shared_rows = value != 0;
    else if (!strcmp(name, "negative_noise_mfma")) g_opt_negative_noise_mfma = value != 0;
    else if (!strcmp(name, "finalize_wave")) g_opt_finalize_wave = value != 0;
    else if (!strcmp(name, "fuse_finalize")) g_opt_fuse_finalize = value != 0;
    else if (!strcmp(name, "pair_models")) g_opt_pair_models = value != 0;
    else if (!strcmp(name, "pack_setup_uploads")) g_opt_pack_setup_uploads = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (!strcmp(name, "pack_transfers")) g_opt_pack_transfers = value != 0;
    else if (!strcmp(name, "null_side_level")) g_opt_null_side_level = value;
    else if (!strcmp(name, "fold_evidence")) g_opt_fold_evidence = value != 0;
    else if (!strcmp(name, "null_side")) g_opt_null_side = value != 0;
    else if (!strcmp(name, "level_spread")) g_opt_level_spread = value != 0;
    else if (!strcmp(name, "level_search_groups")) g_opt_level_search_groups = value < 1 ? 1 : (value > 64 ? 64 : value);
    else return fail(DLAGP_EINVAL, "unknown option '%s'", name);
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_set_option")

int dlagp_plan_cut(int64_t groups, double chunks, int64_t forced_splits, int *full, double *frac) try {
    if (!full || !frac || groups < 1 || !(chunks >= 1.0)) return fail(DLAGP_EINVAL, "bad argument");
    plan::Options opt = planner_options();
    opt.pixel_splits = forced_splits;
    plan::choose_cut((long)groups, chunks, opt, full, frac);
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_plan_cut")

int dlagp_plan_debug(int nb, const int *n_u, const uint8_t *const *keep, int ntaps, int64_t S, const int64_t *options,
                     int64_t *summary, int32_t *slot_pixel, int32_t *slot_pos, int64_t slots_cap, int32_t *passes,
                     int64_t passes_cap, int32_t *items, int64_t items_cap, uint8_t *counts, int64_t counts_cap) try {
    std::string err;
    const int rc = plan::debug(nb, n_u, keep, ntaps, S, options, summary, slot_pixel, slot_pos, slots_cap, passes,
                               passes_cap, items, items_cap, counts, counts_cap, err);
    return rc ? fail(rc, "%s", err.c_str()) : DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_plan_debug")

int dlagp_get_option(const char *name, int64_t *value) try {
    if (!name || !value) return fail(DLAGP_EINVAL, "NULL argument");
    if (!strcmp(name, "pixel_splits")) *value = g_opt_pixel_splits;
    else if (!strcmp(name, "path")) *value = g_opt_path;
    else if (!strcmp(name, "timing")) *value = g_opt_timing;
    else if (!strcmp(name, "debug")) *value = g_opt_debug;
    else if (!strcmp(name, "core_patch")) *value = g_opt_core_patch;
    else if (!strcmp(name, "pool_mb")) *value = g_opt_pool_mb;
    else if (!strcmp(name, "profile_cache")) *value = g_opt_profile_cache;
    else if (!strcmp(name, "profile_cache_max_mb")) *value = g_opt_profile_cache_max_mb;
    else if (!strcmp(name, "plan_item_cost")) *value = g_opt_plan_item_cost;
    else if (!strcmp(name, "plan_rest_first")) *value = g_opt_plan_rest_first;
    else if (!strcmp(name, "plan_slots")) *value = g_opt_plan_slots;
    else if (!strcmp(name, "plan_xcd_affine")) *value = g_opt_plan_xcd_affine;
    else if (!strcmp(name, "plan_taper")) *value = g_opt_plan_taper;
    else if (!strcmp(name, "plan_affine_min_chunks")) *value = g_opt_plan_affine_min_chunks;
    else if (!strcmp(name, "finalize_row16_blocks")) *value = g_opt_finalize_row16_blocks;
    else if (!strcmp(name, "spare_workspaces")) *value = g_opt_spare_workspaces;
    else if (!strcmp(name, "shared_work_stream")) *value = g_opt_shared_work_stream;
    else if (!strcmp(name, "shared_rows")) *value = g_opt_shared_rows;
    else if (!strcmp(name, "negative_noise_mfma")) *value = g_opt_negative_noise_mfma;
    else if (!strcmp(name, "finalize_wave")) *value = g_opt_finalize_wave;
    else if (!strcmp(name, "fuse_finalize")) *value = g_opt_fuse_finalize;
    else if (!strcmp(name, "last_launch_fused")) *value = g_last_launch_fused;
    else if (!strcmp(name, "pair_models")) *value = g_opt_pair_models;
    else if (!strcmp(name, "last_sweep_paired")) *value = g_last_sweep_paired;
    else if (!strcmp(name, "last_evidence_folded")) *value = g_last_evidence_folded;
    else if (!strcmp(name, "fold_evidence")) *value = g_opt_fold_evidence;
    else if (!strcmp(name, "pack_setup_uploads")) *value = g_opt_pack_setup_uploads;
    else if (!strcmp(name, "pack_transfers")) *value = g_opt_pack_transfers;
    else if (!strcmp(name, "null_side_level")) *value = g_opt_null_side_level;
    else if (!strcmp(name, "null_side")) *value = g_opt_null_side;
    else if (!strcmp(name, "level_spread")) *value = g_opt_level_spread;
    else if (!strcmp(name, "level_search_groups")) *value = g_opt_level_search_groups;
    else if (!strcmp(name, "last_level_spread")) *value = g_last_level_spread;
    else if (!strcmp(name, "big_allocs")) *value = g_stat_big_allocs;
    else if (!strcmp(name, "big_frees")) *value = g_stat_big_frees;
    else if (!strcmp(name, "last_launch_path")) *value = g_last_launch_path;
    else if (!strcmp(name, "last_launch_items")) *value = g_last_launch_items;
    else if (!strcmp(name, "last_launch_parts")) *value = g_last_launch_parts;
    else if (!strcmp(name, "last_sweep_cached")) *value = g_last_sweep_cached;
    else if (!strcmp(name, "last_launch_patch")) *value = g_last_launch_patch;
    else if (!strcmp(name, "last_launch_groups")) *value = g_last_launch_groups;
    else if (!strcmp(name, "work_stream_idle")) {
        // diagnostics (tools/measure_catalogue.py, DLAGP_POLL_IDLE): 1 when the current device's shared kernel stream has
        // nothing queued or running -- polled from a side thread it says how often the host leaves the device without work
        // (asks without side effects: no stream is created for the question, and the "not ready" answer of a busy stream is
        // not left behind as the polling thread's last HIP error)
        int device = 0;
        HIP_TRY(hipGetDevice(&device));
        hipStream_t shared = shared_stream_if_any(device);
        if (!g_opt_shared_work_stream || !shared) *value = 0;
        else {
            *value = hipStreamQuery(shared) == hipSuccess;
            (void)hipGetLastError();
        }
    }
    else return fail(DLAGP_EINVAL, "unknown option '%s'", name);
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_get_option")

int dlagp_timing_read(double *total_ms, int64_t *launches, int reset) try {
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    double total = 0.0;
    int64_t counted = 0;
    for (const EventPair &ev : g_timing_events) {
        // a pair that cannot be read (its stream was destroyed, a sticky device error) is skipped, never fatal:
        // the list is still cleared below, so one bad pair cannot wedge every later read
        float ms = 0.f;
        if (hipEventSynchronize(ev.stop) != hipSuccess || hipEventElapsedTime(&ms, ev.start, ev.stop) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        total += ms;
        ++counted;
    }
    if (total_ms) *total_ms = total;
    if (launches) *launches = counted;
    if (reset) {
        for (const EventPair &ev : g_timing_events) {
            (void)hipEventDestroy(ev.start);
            (void)hipEventDestroy(ev.stop);
        }
        g_timing_events.clear();
    }
    return DLAGP_OK;
} DLAGP_CATCH_INT("dlagp_timing_read")

}  // extern "C"
