// TEST INFRASTRUCTURE: the host runtime of libdlagp_hip.so -- the whole host side of dlagp_hip.hip, through its C ABI --
// over a fake device (fake_device.h), driven by several host threads in the catalogue driver's call pattern.  Built as an
// executable with -fsanitize=thread and with -fsanitize=address,undefined (`make sanitize`), run by
// tests/test_runtime_sanitizers.py on the CPU.  What it exercises: the per-thread stream / arena / lane pools, the device
// buffer pool and the spare workspaces, batch membership, deferred context set-up used across threads (and by two batches
// at once), the two-phase model-evidence call and its cancellation, destruction orders, allocation failures in the middle
// of a batch, asynchronous device-pointer calls drained by destroy.  Exit code 0 = no error code where none was expected, no
// buffer recycled in flight, nothing leaked; the sanitizers add their own verdict.
//
//   runtime_replay [batches per producer = 24] [seed = 1]
#define DLAGP_FAKE_DEVICE 1
#include "dlagp_hip.hip"

#include <condition_variable>
#include <deque>
#include <random>
#include <thread>

namespace {

struct Batch {
    std::vector<dlagp_ctx *> ctxs;
    dlagp_batch *batch = nullptr;
    int id = 0;
};

template <typename T>
struct Queue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<T> items;
    bool closed = false;
    void push(T v) {
        {
            std::lock_guard<std::mutex> lock(m);
            items.push_back(std::move(v));
        }
        cv.notify_one();
    }
    bool pop(T &out) {
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return !items.empty() || closed; });
        if (items.empty()) return false;
        out = std::move(items.front());
        items.pop_front();
        return true;
    }
    void close() {
        {
            std::lock_guard<std::mutex> lock(m);
            closed = true;
        }
        cv.notify_all();
    }
};

std::atomic<long> g_unexpected{0}, g_injected_failures{0}, g_calls{0};

// the only failures expected are the injected allocation failures (DLAGP_EDEVICE from the fake hipMalloc)
bool ok(int rc, const char *what, bool may_fail) {
    g_calls.fetch_add(1);
    // the kernels of this build do nothing: the non-PD flag a real accumulate launch clears holds whatever the recycled buffer
    // held before, so "not positive definite" is no finding here (nothing numerical is)
    if (rc == DLAGP_OK || rc == DLAGP_ELINALG) return true;
    // an injected allocation failure says so in its message (fake hipMalloc): expected whenever it comes, whichever thread's
    // call it lands in and however long that call takes
    (void)may_fail;
    if (rc == DLAGP_EDEVICE && strstr(dlagp_last_error(), "(fake)")) {
        g_injected_failures.fetch_add(1);
        return false;
    }
    std::fprintf(stderr, "runtime_replay: %s returned %d: %s\n", what, rc, dlagp_last_error());
    g_unexpected.fetch_add(1);
    return false;
}

struct Spectrum {
    int n, n_u;
    std::vector<double> x, y, v, lam_kept, padded;
    std::vector<uint8_t> keep;
    double z_qso;
};

Spectrum make_spectrum(std::mt19937 &rng, int n_u) {
    Spectrum s;
    s.n_u = n_u;
    s.z_qso = 2.5;
    s.keep.assign(n_u, 1);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    for (int j = 0; j < n_u; ++j)
        if (uni(rng) < 0.05) s.keep[j] = 0;
    s.keep[0] = s.keep[n_u - 1] = 1;
    s.padded.resize(n_u + 6);
    for (int e = 0; e < n_u + 6; ++e) s.padded[e] = 3700.0 * std::pow(10.0, 1e-4 * (e - 3));
    for (int j = 0; j < n_u; ++j)
        if (s.keep[j]) {
            const double lam = s.padded[j + 3];
            s.lam_kept.push_back(lam);
            s.x.push_back(lam / (1.0 + s.z_qso));
            s.y.push_back(1.0 + 0.1 * uni(rng));
            s.v.push_back(0.01 + 0.01 * uni(rng));
        }
    s.n = (int)s.x.size();
    return s;
}

// a RAW spectrum for dlagp_batch_prepare: observed wavelengths whose rest frame covers [1005, 1199] A of the model grid
struct RawSpectrum {
    std::vector<double> wavelengths, flux, noise;
    std::vector<uint8_t> mask;
    double z_qso;
};
RawSpectrum make_raw(std::mt19937 &rng, int num) {
    RawSpectrum s;
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    s.z_qso = 2.2 + 0.6 * uni(rng);
    const double step = std::log10(1199.0 / 1005.0) / (num - 1);
    for (int j = 0; j < num; ++j) {
        s.wavelengths.push_back(1005.0 * std::pow(10.0, step * j) * (1.0 + s.z_qso));
        s.flux.push_back(1.0 + 0.2 * uni(rng));
        s.noise.push_back(0.01 + 0.01 * uni(rng));
        s.mask.push_back(uni(rng) < 0.05 ? 1 : 0);
    }
    return s;
}
const dlagp_prepare_params kPrepare = {1010.0, 1190.0, 1191.0, 1199.0, 1e-4, 1215.6701, 911.7633, 0.01, 0.01, 3, 1};

}  // namespace

// The checker must have teeth: a buffer handed back while a launch on an unsynchronised stream still uses it, a double
// free and a second destroy of an event's owner-side handle are each reported by the fake device.
static int selftest() {
    long before = fake::error_count();
    hipStream_t stream = nullptr;
    (void)hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    double *p = nullptr;
    (void)hipMalloc((void **)&p, 1024);
    hipLaunchKernelGGL(nothing, dim3(1), dim3(64), 0, stream, p, 3);
    const bool flagged_in_flight = !fake::check_idle(p, "selftest") && fake::error_count() == before + 1;
    (void)hipStreamSynchronize(stream);
    const bool idle_after_sync = fake::check_idle(p, "selftest");
    // the shared kernel stream: a call's kernels run on `stream`, its copies on `copies` behind an event, and the call
    // waits for `copies` only.  That wait covers the kernels before the event -- and not the one launched after it.
    hipStream_t copies = nullptr;
    hipEvent_t computed = nullptr;
    (void)hipStreamCreateWithFlags(&copies, hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&computed, hipEventDisableTiming);
    double *q = nullptr;
    (void)hipMalloc((void **)&q, 1024);
    hipLaunchKernelGGL(nothing, dim3(1), dim3(64), 0, stream, p, 3);
    (void)hipEventRecord(computed, stream);
    hipLaunchKernelGGL(nothing, dim3(1), dim3(64), 0, stream, q, 3);
    (void)hipStreamWaitEvent(copies, computed, 0);
    (void)hipStreamSynchronize(copies);
    before = fake::error_count();
    const bool wait_covers_event = fake::check_idle(p, "selftest") && fake::error_count() == before;
    const bool wait_stops_at_event = !fake::check_idle(q, "selftest") && fake::error_count() == before + 1;
    (void)hipStreamSynchronize(stream);
    (void)hipFree(q);
    (void)hipEventDestroy(computed);
    (void)hipFree(p);
    before = fake::error_count();
    const bool flagged_double_free = hipFree(p) != hipSuccess && fake::error_count() == before + 1;
    std::printf("{\"selftest\": true, \"in_flight_recycle_flagged\": %s, \"idle_after_sync\": %s, \"double_free_flagged\": %s, "
                "\"stream_wait_covers_event\": %s, \"stream_wait_stops_at_event\": %s}\n",
                flagged_in_flight ? "true" : "false", idle_after_sync ? "true" : "false", flagged_double_free ? "true" : "false",
                wait_covers_event ? "true" : "false", wait_stops_at_event ? "true" : "false");
    return flagged_in_flight && idle_after_sync && flagged_double_free && wait_covers_event && wait_stops_at_event ? 0 : 1;
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "--selftest")) return selftest();
    const int per_producer = argc > 1 ? atoi(argv[1]) : 24;
    const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1u;
    const int K = 20, S = 96, MAX_DLAS = 3, NB = 4;
    // odd seeds: every thread's kernels on the one kernel stream of the device (the default); even seeds: on the
    // thread's own stream
    if (dlagp_set_option("shared_work_stream", seed % 2) != DLAGP_OK) return 2;

    // the learned model: a grid that covers every spectrum's rest wavelengths
    const int n_grid = 400;
    std::vector<double> grid(n_grid), mu(n_grid, 1.0), log_omega(n_grid, -2.0), M((size_t)n_grid * K, 0.01);
    for (int i = 0; i < n_grid; ++i) grid[i] = 1000.0 + 0.5 * i;
    dlagp_model *model = nullptr;
    if (!ok(dlagp_model_create(&model, n_grid, K, grid.data(), mu.data(), M.data(), K, 1, log_omega.data(), -2.3, -5.9, 1.2,
                               0.0023, 3.65, 31, 0), "dlagp_model_create", false))
        return 2;

    std::vector<double> sample_z(NB * S), sub_z(NB * S), nhi(S), sub_nhi(S), uniforms((size_t)NB * (MAX_DLAS - 1) * S);
    std::mt19937 rng0(seed);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    for (auto &z : sample_z) z = 2.1 + 0.3 * uni(rng0);
    sub_z = sample_z;
    for (int i = 0; i < S; ++i) {
        nhi[i] = std::pow(10.0, 20.0 + 3.0 * uni(rng0));
        sub_nhi[i] = std::pow(10.0, 19.5 + 0.5 * uni(rng0));
    }
    for (auto &u : uniforms) u = uni(rng0);

    Queue<Batch> prepared, swept;
    std::atomic<int> next_id{0};
    std::atomic<bool> inject{false};

    // -- prepare threads: contexts created WITHOUT waiting for their set-up (the first use waits, on another thread) --
    auto producer = [&](unsigned thread_seed) {
        std::mt19937 rng(thread_seed);
        for (int b = 0; b < per_producer; ++b) {
            Batch batch;
            batch.id = next_id.fetch_add(1);
            bool failed = false;
            if (batch.id % 5 == 2) {  // the catalogue driver's route since round 6: raw spectra, ONE call for the whole batch
                std::vector<RawSpectrum> raw;
                for (int i = 0; i < NB; ++i) raw.push_back(make_raw(rng, 300 + (int)(rng() % 400)));
                std::vector<int> num;
                std::vector<const double *> w, f, nv;
                std::vector<const uint8_t *> m;
                std::vector<double> zq;
                for (const RawSpectrum &r : raw) {
                    num.push_back((int)r.wavelengths.size());
                    w.push_back(r.wavelengths.data());
                    f.push_back(r.flux.data());
                    nv.push_back(r.noise.data());
                    m.push_back(r.mask.data());
                    zq.push_back(r.z_qso);
                }
                std::vector<double> offsets(S), tables((size_t)NB * S), scalars((size_t)NB * 8);
                for (int i = 0; i < S; ++i) offsets[i] = (i + 0.5) / S;
                batch.ctxs.assign(NB, nullptr);
                const int rc = dlagp_batch_prepare(&batch.batch, batch.ctxs.data(), model, &kPrepare, NB, num.data(), w.data(), f.data(),
                                                   nv.data(), m.data(), zq.data(), 3, DLAGP_FLAG_BROADENING, nullptr, nullptr, S,
                                                   offsets.data(), batch.id % 2 ? offsets.data() : nullptr, tables.data(),
                                                   batch.id % 2 ? tables.data() : nullptr, scalars.data());
                if (!ok(rc, "dlagp_batch_prepare", inject.load())) {
                    for (dlagp_ctx *c : batch.ctxs)  // a failed call leaves nothing behind
                        if (c) {
                            std::fprintf(stderr, "runtime_replay: a failed dlagp_batch_prepare left a context behind\n");
                            g_unexpected.fetch_add(1);
                        }
                    continue;
                }
                if (!(scalars[1] < scalars[2]) || scalars[5] < 1.0) {
                    std::fprintf(stderr, "runtime_replay: dlagp_batch_prepare returned nonsense bounds\n");
                    g_unexpected.fetch_add(1);
                }
                prepared.push(std::move(batch));
                continue;
            }
            for (int i = 0; i < NB && !failed; ++i) {
                Spectrum s = make_spectrum(rng, 120 + (int)(rng() % 200));
                dlagp_ctx *ctx = nullptr;
                const int flags = DLAGP_FLAG_BROADENING | ((batch.id % 3) ? DLAGP_FLAG_DEFER_SYNC : 0);
                const bool may_fail = inject.load();
                int rc;
                if (batch.id % 5 == 4) {  // the host-array route (dlagp_create), rank 12
                    std::vector<double> m((size_t)s.n * 12, 0.02), om(s.n, 0.01);
                    rc = dlagp_create(&ctx, s.n, s.n_u, 12, s.y.data(), s.y.data(), m.data(), 12, 1, om.data(), s.v.data(),
                                      s.padded.data(), s.keep.data(), 3, DLAGP_FLAG_BROADENING, 0);
                } else {
                    rc = dlagp_create_from_model(&ctx, model, s.n, s.n_u, s.x.data(), s.y.data(), s.v.data(), s.lam_kept.data(),
                                                 s.z_qso, s.padded.data(), s.keep.data(), 3, flags);
                }
                if (!ok(rc, "context creation", may_fail || inject.load())) {
                    failed = true;
                    break;
                }
                batch.ctxs.push_back(ctx);
            }
            if (failed) {  // the driver's failure path: everything prepared so far is released
                for (dlagp_ctx *c : batch.ctxs) dlagp_destroy(c);
                continue;
            }
            prepared.push(std::move(batch));
        }
    };

    // -- sweep threads: batch from contexts another thread created; the two-phase call; asynchronous device calls --
    auto sweeper = [&](unsigned thread_seed) {
        std::mt19937 rng(thread_seed);
        Batch b;
        std::vector<double> nulls(NB), sub_ev(NB), sub_ll((size_t)NB * S), ev((size_t)NB * MAX_DLAS), ll((size_t)NB * MAX_DLAS * S);
        std::vector<int32_t> base((size_t)NB * (MAX_DLAS - 1) * S);
        while (prepared.pop(b)) {
            const bool may_fail = inject.load();
            const bool mixed_rank = false;
            (void)mixed_rank;
            int rc = b.batch ? DLAGP_OK : dlagp_batch_create(&b.batch, b.ctxs.data(), (int)b.ctxs.size());  // (dlagp_batch_prepare made its own)
            if (!ok(rc, "dlagp_batch_create", may_fail || inject.load())) {
                b.batch = nullptr;
                swept.push(std::move(b));
                continue;
            }
            const int nb = (int)b.ctxs.size();
            switch (rng() % 5) {
            case 0: {  // the catalogue driver's call
                rc = dlagp_batch_model_evidences(b.batch, S, MAX_DLAS, sub_z.data(), sub_nhi.data(), 0, sample_z.data(), nhi.data(), 0,
                                                 0.01, uniforms.data(), nulls.data(), sub_ev.data(), sub_ll.data(), ev.data(), ll.data(),
                                                 base.data());
                if (ok(rc, "dlagp_batch_model_evidences", may_fail || inject.load())) {
                    std::vector<int32_t> ties((size_t)nb * (MAX_DLAS - 1));
                    ok(dlagp_batch_resample_ties(b.batch, ties.data(), (int64_t)ties.size()), "dlagp_batch_resample_ties", false);
                }
                break;
            }
            case 1: {  // two phases, host work in between
                rc = dlagp_batch_model_evidences_start(b.batch, S, MAX_DLAS, sub_z.data(), sub_nhi.data(), 0, sample_z.data(), nhi.data(),
                                                       0, 0.01);
                if (ok(rc, "dlagp_batch_model_evidences_start", may_fail || inject.load())) {
                    std::this_thread::yield();
                    rc = dlagp_batch_model_evidences_finish(b.batch, uniforms.data(), nulls.data(), sub_ev.data(), sub_ll.data(), ev.data(),
                                                            ll.data(), base.data());
                    ok(rc, "dlagp_batch_model_evidences_finish", may_fail || inject.load());
                }
                break;
            }
            case 2: {  // a started call that is never finished: cancelled by the next call on the batch
                rc = dlagp_batch_model_evidences_start(b.batch, S, MAX_DLAS, sub_z.data(), sub_nhi.data(), 0, sample_z.data(), nhi.data(),
                                                       0, 0.01);
                if (ok(rc, "dlagp_batch_model_evidences_start", may_fail || inject.load())) {
                    ok(dlagp_batch_null(b.batch, nulls.data()), "dlagp_batch_null", may_fail || inject.load());
                    rc = dlagp_batch_model_evidences_finish(b.batch, uniforms.data(), nulls.data(), sub_ev.data(), sub_ll.data(), ev.data(),
                                                            ll.data(), base.data());
                    if (rc != DLAGP_EINVAL) {
                        std::fprintf(stderr, "runtime_replay: _finish after a cancelled _start returned %d, expected DLAGP_EINVAL\n", rc);
                        g_unexpected.fetch_add(1);
                    }
                    ok(dlagp_batch_model_evidences_cancel(b.batch), "dlagp_batch_model_evidences_cancel", false);
                }
                break;
            }
            case 3: {  // host-pointer sweeps and evaluations, growing sample counts (scratch growth)
                for (int64_t s_now : {(int64_t)17, (int64_t)S}) {
                    rc = dlagp_batch_sweep(b.batch, s_now, 2, sample_z.data(), nhi.data(), 0, 0.01, uniforms.data(), ev.data(), ll.data(),
                                           base.data());
                    ok(rc, "dlagp_batch_sweep", may_fail || inject.load());
                }
                std::vector<double> z1((size_t)nb * 33, 2.2), n1((size_t)nb * 33, 1e20), o1((size_t)nb * 33);
                ok(dlagp_batch_eval(b.batch, 33, 1, z1.data(), n1.data(), o1.data()), "dlagp_batch_eval", may_fail || inject.load());
                ok(dlagp_eval(b.ctxs[0], 33, 1, z1.data(), n1.data(), o1.data()), "dlagp_eval", may_fail || inject.load());
                double null_one = 0.0;
                ok(dlagp_null(b.ctxs[nb - 1], &null_one), "dlagp_null", may_fail || inject.load());
                break;
            }
            default: {  // device-pointer calls on a stream of the caller's, left in flight: destroy must drain them
                hipStream_t stream = nullptr;
                (void)hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
                double *d_z = nullptr, *d_nhi = nullptr, *d_ev = nullptr, *d_ll = nullptr, *d_u = nullptr;
                int32_t *d_base = nullptr;
                bool got = hipMalloc((void **)&d_z, sizeof(double) * nb * S) == hipSuccess;
                got = got && hipMalloc((void **)&d_nhi, sizeof(double) * S) == hipSuccess;
                got = got && hipMalloc((void **)&d_ev, sizeof(double) * nb * MAX_DLAS) == hipSuccess;
                got = got && hipMalloc((void **)&d_ll, sizeof(double) * nb * MAX_DLAS * S) == hipSuccess;
                got = got && hipMalloc((void **)&d_u, sizeof(double) * nb * (MAX_DLAS - 1) * S) == hipSuccess;
                got = got && hipMalloc((void **)&d_base, sizeof(int32_t) * nb * (MAX_DLAS - 1) * S) == hipSuccess;
                if (got) {
                    rc = dlagp_batch_sweep_device(b.batch, S, MAX_DLAS, d_z, d_nhi, 0, 0.01, d_u, d_ev, d_ll, d_base, stream);
                    ok(rc, "dlagp_batch_sweep_device", may_fail || inject.load());
                    std::vector<int32_t> ties((size_t)nb * (MAX_DLAS - 1));  // waits for the sweep's stream, then copies
                    if (rc == DLAGP_OK)
                        ok(dlagp_batch_resample_ties(b.batch, ties.data(), (int64_t)ties.size()), "dlagp_batch_resample_ties (device)",
                           may_fail || inject.load());
                    rc = dlagp_batch_eval_device(b.batch, S, 2, d_z, d_z, d_ll, stream);  // kd = 2: [S][2] fits d_ll? no: reads only
                    ok(rc, "dlagp_batch_eval_device", may_fail || inject.load());
                } else {
                    g_injected_failures.fetch_add(1);
                }
                // the caller's own arrays outlive the launches: wait for its stream before freeing them
                (void)hipStreamSynchronize(stream);
                for (void *p : {(void *)d_z, (void *)d_nhi, (void *)d_ev, (void *)d_ll, (void *)d_u, (void *)d_base})
                    if (p) (void)hipFree(p);
                break;
            }
            }
            swept.push(std::move(b));
        }
    };

    // -- finish thread: destruction, in both orders --
    auto finisher = [&]() {
        Batch b;
        while (swept.pop(b)) {
            if (b.id % 4 == 1 && !b.ctxs.empty()) {  // a member destroyed before its batch: the batch refuses further calls
                dlagp_destroy(b.ctxs[0]);
                b.ctxs[0] = nullptr;
                if (b.batch) {
                    double out[8];
                    const int rc = dlagp_batch_null(b.batch, out);
                    if (rc != DLAGP_EINVAL) {
                        std::fprintf(stderr, "runtime_replay: a batch with a destroyed member answered %d, expected DLAGP_EINVAL\n", rc);
                        g_unexpected.fetch_add(1);
                    }
                }
            }
            if (b.batch) dlagp_batch_destroy(b.batch);
            for (dlagp_ctx *c : b.ctxs)
                if (c) dlagp_destroy(c);
        }
    };

    // -- one deferred context, first used by two batches on two threads at once (the set-up event must be waited for and
    //    destroyed exactly once) --
    auto shared_member = [&](unsigned thread_seed) {
        std::mt19937 rng(thread_seed);
        for (int round = 0; round < per_producer / 2 + 1; ++round) {
            Spectrum s = make_spectrum(rng, 150);
            dlagp_ctx *ctx = nullptr;
            if (!ok(dlagp_create_from_model(&ctx, model, s.n, s.n_u, s.x.data(), s.y.data(), s.v.data(), s.lam_kept.data(), s.z_qso,
                                            s.padded.data(), s.keep.data(), 3, DLAGP_FLAG_BROADENING | DLAGP_FLAG_DEFER_SYNC),
                    "shared member creation", inject.load()))
                continue;
            dlagp_batch *one = nullptr, *two = nullptr;
            std::thread other([&] { ok(dlagp_batch_create(&two, &ctx, 1), "dlagp_batch_create (shared member)", inject.load()); });
            ok(dlagp_batch_create(&one, &ctx, 1), "dlagp_batch_create (shared member)", inject.load());
            other.join();
            if (one) dlagp_batch_destroy(one);
            if (two) dlagp_batch_destroy(two);
            dlagp_destroy(ctx);
        }
    };

    // -- failure injection: every so often one of the next few device allocations fails, on whichever thread makes it --
    std::atomic<bool> stop{false};
    auto saboteur = [&]() {
        std::mt19937 rng(seed * 7919u + 13u);
        while (!stop.load()) {
            std::this_thread::sleep_for(std::chrono::milliseconds(3));
            inject.store(true);
            fake::fail_allocations_after(1 + (long)(rng() % 40));
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            fake::fail_allocations_after(-1);
            // calls that began under injection may still be running: the flag stays up a little longer than the injection
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
            inject.store(false);
            // every few rounds the pools are emptied under the other threads' feet (idle buffers only: that is the pool's
            // contract), so that the steady state keeps allocating and the injected failures find something to fail
            if (rng() % 3 == 0) {
                (void)dlagp_set_option("pool_mb", 0);
                (void)dlagp_set_option("pool_mb", 2048);
            }
        }
        fake::fail_allocations_after(-1);
    };

    std::thread p1(producer, seed * 101u + 1u), p2(producer, seed * 101u + 2u);
    std::thread s1(sweeper, seed * 211u + 1u), s2(sweeper, seed * 211u + 2u), s3(sweeper, seed * 211u + 3u);
    std::thread f1(finisher), sh(shared_member, seed * 307u), sab(saboteur);
    p1.join();
    p2.join();
    prepared.close();
    s1.join();
    s2.join();
    s3.join();
    swept.close();
    f1.join();
    sh.join();
    stop.store(true);
    sab.join();

    dlagp_model_destroy(model);
    // everything parked in the pools goes back to the (fake) allocator: whatever is left is a leak
    ok(dlagp_set_option("pool_mb", 0), "dlagp_set_option(pool_mb, 0)", false);
    size_t leaked = 0;
    {
        fake::State &st = fake::state();
        std::lock_guard<std::mutex> lock(st.mutex);
        leaked = st.allocations.size();
        for (auto &kv : st.allocations)
            if (leaked <= 8) std::fprintf(stderr, "runtime_replay: leaked device allocation of %zu bytes\n", kv.second.bytes);
    }
    std::printf("{\"calls\": %ld, \"launches\": %ld, \"device_mallocs\": %ld, \"device_frees\": %ld, \"injected_failures\": %ld, "
                "\"unexpected_errors\": %ld, \"fake_device_errors\": %ld, \"leaked_allocations\": %zu}\n",
                g_calls.load(), fake::state().launches.load(), fake::state().mallocs.load(), fake::state().frees.load(),
                g_injected_failures.load(), g_unexpected.load(), fake::error_count(), leaked);
    return (g_unexpected.load() || fake::error_count() || leaked) ? 1 : 0;
}
