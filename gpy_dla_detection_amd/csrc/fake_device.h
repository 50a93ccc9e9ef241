// TEST INFRASTRUCTURE, never part of libdlagp_hip.so: a host-only stand-in for the subset of the HIP runtime API that the
// host side of the library (dlagp_hip.hip) calls, so that THAT code -- buffer pool, stream / arena / lane pools, spare
// workspaces, batch membership, the deferred-set-up state machine, staging of every transfer, launch bookkeeping -- can
// be compiled with g++ and run under ThreadSanitizer and AddressSanitizer + UBSan on the CPU (runtime_replay.cpp,
// `make sanitize`, tests/test_runtime_sanitizers.py).  GPU sanitizers are not available on the pool; the host runtime is
// where the round-3 advisor found real thread bugs.
//
// What the fake keeps of the real semantics, because the host code's correctness rests on it:
//   * "device" memory is host memory (calloc): every hipMemcpy really copies, so a wrong size or offset in the staging
//     logic is an ASan report; double frees and frees of foreign pointers are detected by the registry;
//   * streams are counters: every operation enqueued on a stream gets a ticket, and only a synchronisation of that
//     stream (or of an event recorded on it, or of the device) completes tickets.  A launch marks the allocations its
//     pointer arguments lie in with its (stream, ticket);
//   * fake::check_idle(ptr) -- called where the library hands a buffer back to its pool (DLAGP_DEVICE_CHECK_IDLE in
//     DeviceBuffer::release) and by hipFree -- fails when the allocation still carries an uncompleted ticket: the library
//     would be recycling memory that a kernel in flight reads or writes ("destroy and scratch growth drain the stream",
//     include/dlagp.h);
//   * fake::fail_allocations_after(n) makes the n-th following hipMalloc fail: the failure paths (a batch that cannot be
//     built, a context whose arrays cannot be allocated) run under the sanitizers too.
// Kernels do nothing.  Nothing numerical is checked here -- that is the GPU tests' job.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#if defined(__has_include)
#if __has_include(<sanitizer/common_interface_defs.h>)
#include <sanitizer/common_interface_defs.h>
#define DLAGP_FAKE_PRINT_STACK() __sanitizer_print_stack_trace()
#endif
#endif
#ifndef DLAGP_FAKE_PRINT_STACK
#define DLAGP_FAKE_PRINT_STACK() ((void)0)
#endif

enum hipError_t { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorInvalidDevice = 101, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace fake {

struct Stream;
struct Wait {  // hipStreamWaitEvent: the waiting stream's operations from ticket `at` on run after (stream, ticket)
    uint64_t at;
    Stream *stream;
    uint64_t ticket;
};
struct Stream {
    int device = 0;
    std::atomic<uint64_t> issued{0}, completed{0};
    std::vector<Wait> waits;  // guarded by State::mutex
};
struct Event {
    Stream *stream = nullptr;
    uint64_t ticket = 0;
    bool alive = true;
};
struct Allocation {
    size_t bytes = 0;
    std::vector<std::pair<Stream *, uint64_t>> marks;  // launches that may still touch it
};

struct State {
    std::mutex mutex;
    std::map<char *, Allocation> allocations;  // by start address
    std::vector<Stream *> streams;
    std::map<int, Stream *> null_streams;
    std::atomic<long> mallocs_until_failure{-1};
    std::atomic<long> errors{0};
    std::atomic<long> launches{0}, mallocs{0}, frees{0};
};
inline State &state() {
    static State *s = new State();  // never destroyed: thread_local destructors of the library may run late
    return *s;
}
inline void report(const char *what, const void *ptr) {
    state().errors.fetch_add(1);
    std::fprintf(stderr, "fake_device: %s (%p)\n", what, ptr);
}
inline thread_local int current_device = 0;

inline Stream *resolve(Stream *s) {
    if (s) return s;
    State &st = state();
    std::lock_guard<std::mutex> lock(st.mutex);
    Stream *&null_stream = st.null_streams[current_device];
    if (!null_stream) {
        null_stream = new Stream();
        null_stream->device = current_device;
        st.streams.push_back(null_stream);
    }
    return null_stream;
}
inline uint64_t enqueue(Stream *s) { return resolve(s)->issued.fetch_add(1) + 1; }
// Everything on `s` up to ticket `upto` has run -- and with it whatever those operations had waited for on other streams
// (hipStreamWaitEvent): a call of the library that drains its transfer stream has thereby seen its kernels on the shared
// kernel stream complete.  Call with State::mutex held.
inline void complete_locked(Stream *s, uint64_t upto, int depth = 0) {
    uint64_t seen = s->completed.load();
    while (seen < upto && !s->completed.compare_exchange_weak(seen, upto)) {
    }
    if (depth > 8) return;
    for (size_t i = 0; i < s->waits.size();) {
        const Wait w = s->waits[i];
        if (w.at <= upto) {
            s->waits[i] = s->waits.back();
            s->waits.pop_back();
            complete_locked(w.stream, w.ticket, depth + 1);
        } else {
            ++i;
        }
    }
}
inline void complete(Stream *s, uint64_t upto) {
    std::lock_guard<std::mutex> lock(state().mutex);
    complete_locked(s, upto);
}

// the allocation that holds `p`, or end()
inline std::map<char *, Allocation>::iterator find_locked(State &st, const void *p) {
    auto it = st.allocations.upper_bound((char *)p);
    if (it == st.allocations.begin()) return st.allocations.end();
    --it;
    return (char *)p < it->first + it->second.bytes ? it : st.allocations.end();
}
inline void mark(const void *p, Stream *s, uint64_t ticket) {
    if (!p) return;
    State &st = state();
    std::lock_guard<std::mutex> lock(st.mutex);
    auto it = find_locked(st, p);
    if (it == st.allocations.end()) return;  // a host pointer, or memory of someone else
    auto &marks = it->second.marks;
    marks.erase(std::remove_if(marks.begin(), marks.end(), [](const std::pair<Stream *, uint64_t> &m) { return m.first->completed.load() >= m.second; }),
                marks.end());
    marks.push_back({s, ticket});
}
// fails (and counts an error) when a launch that touched p's allocation has not been waited for
inline bool check_idle(const void *p, const char *where) {
    if (!p) return true;
    State &st = state();
    std::lock_guard<std::mutex> lock(st.mutex);
    auto it = find_locked(st, p);
    if (it == st.allocations.end()) {
        report("check_idle on memory the fake device did not allocate", p);
        return false;
    }
    for (const auto &m : it->second.marks)
        if (m.first->completed.load() < m.second) {
            std::fprintf(stderr, "fake_device: %s: buffer %p recycled while launch %llu of stream %p is in flight (completed %llu)\n",
                         where, p, (unsigned long long)m.second, (void *)m.first, (unsigned long long)m.first->completed.load());
            st.errors.fetch_add(1);
            DLAGP_FAKE_PRINT_STACK();  // who recycles it
            return false;
        }
    it->second.marks.clear();
    return true;
}
inline void fail_allocations_after(long n) { state().mallocs_until_failure.store(n); }
inline long error_count() { return state().errors.load(); }

// what a launch may touch: every pointer argument, and the pointers inside the argument structs the kernels take by value
template <typename T>
inline void mark_arg(const T &, Stream *, uint64_t) {}
template <typename T>
inline void mark_arg(T *const &p, Stream *s, uint64_t t) {
    mark((const void *)p, s, t);
}

}  // namespace fake

typedef fake::Stream *hipStream_t;
typedef fake::Event *hipEvent_t;

inline const char *hipGetErrorString(hipError_t e) {
    return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory (fake)" : "error (fake)";
}
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *count) {
    const char *env = getenv("DLAGP_FAKE_DEVICES");
    *count = env ? atoi(env) : 1;
    return hipSuccess;
}
inline hipError_t hipSetDevice(int device) {
    fake::current_device = device;
    return hipSuccess;
}
inline hipError_t hipGetDevice(int *device) {
    *device = fake::current_device;
    return hipSuccess;
}
inline hipError_t hipMalloc(void **out, size_t bytes) {
    fake::State &st = fake::state();
    long left = st.mallocs_until_failure.load();
    while (left >= 0 && !st.mallocs_until_failure.compare_exchange_weak(left, left - 1)) {
    }
    if (left == 0) {
        *out = nullptr;
        return hipErrorOutOfMemory;
    }
    // the sweep workspaces of the real sizes are gigabytes: nothing here computes, so a capped allocation stands in for
    // anything above 64 MB (copies into it are bounds-checked against the REQUESTED size by the registry)
    const size_t real = std::min<size_t>(std::max<size_t>(bytes, 1), (size_t)64 << 20);
    char *p = (char *)calloc(real, 1);
    if (!p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lock(st.mutex);
    st.allocations[p].bytes = real;
    st.mallocs.fetch_add(1);
    *out = p;
    return hipSuccess;
}
inline hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    fake::State &st = fake::state();
    {  // the real hipFree synchronises the device
        std::lock_guard<std::mutex> lock(st.mutex);
        for (fake::Stream *s : st.streams) fake::complete_locked(s, s->issued.load());
        auto it = st.allocations.find((char *)p);
        if (it == st.allocations.end()) {
            fake::report("hipFree of a pointer that is not a live allocation (double free?)", p);
            return hipErrorInvalidValue;
        }
        st.allocations.erase(it);
        st.frees.fetch_add(1);
    }
    free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void **out, size_t bytes, unsigned) {
    *out = malloc(std::max<size_t>(bytes, 1));
    return *out ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipHostFree(void *p) {
    free(p);
    return hipSuccess;
}
inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
    *free_b = (size_t)200 << 30;
    *total_b = (size_t)288 << 30;
    return hipSuccess;
}
namespace fake {
// the device side of a copy must lie inside one live allocation (capped allocations: the copy is clipped to the stand-in)
inline size_t device_room(const void *p) {
    State &st = state();
    std::lock_guard<std::mutex> lock(st.mutex);
    auto it = find_locked(st, p);
    if (it == st.allocations.end()) {
        report("copy to / from device memory that is not a live allocation", p);
        return 0;
    }
    return (size_t)(it->first + it->second.bytes - (char *)p);
}
}  // namespace fake
inline hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    const size_t room = fake::device_room(kind == hipMemcpyHostToDevice ? dst : src);
    memcpy(dst, src, std::min(bytes, room));
    return room ? hipSuccess : hipErrorInvalidValue;
}
inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t stream) {
    fake::Stream *s = fake::resolve(stream);
    const uint64_t ticket = fake::enqueue(s);
    fake::mark(kind == hipMemcpyHostToDevice ? dst : src, s, ticket);
    return hipMemcpy(dst, src, bytes, kind);
}
inline hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t stream) {
    fake::Stream *s = fake::resolve(stream);
    fake::mark(dst, s, fake::enqueue(s));
    const size_t room = fake::device_room(dst);
    memset(dst, value, std::min(bytes, room));
    return room ? hipSuccess : hipErrorInvalidValue;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *out, unsigned) {
    fake::Stream *s = new fake::Stream();
    s->device = fake::current_device;
    fake::State &st = fake::state();
    std::lock_guard<std::mutex> lock(st.mutex);
    st.streams.push_back(s);
    *out = s;
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithPriority(hipStream_t *out, unsigned flags, int) { return hipStreamCreateWithFlags(out, flags); }
inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) {
    *least = 0;
    *greatest = -1;
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t stream) {
    fake::Stream *s = fake::resolve(stream);
    fake::complete(s, s->issued.load());
    return hipSuccess;
}
inline hipError_t hipStreamQuery(hipStream_t stream) {
    fake::Stream *s = fake::resolve(stream);
    return s->completed.load() >= s->issued.load() ? hipSuccess : hipErrorNotReady;
}
inline hipError_t hipDeviceSynchronize() {
    fake::State &st = fake::state();
    std::lock_guard<std::mutex> lock(st.mutex);
    for (fake::Stream *s : st.streams) fake::complete_locked(s, s->issued.load());
    return hipSuccess;
}
inline hipError_t hipEventCreateWithFlags(hipEvent_t *out, unsigned) {
    *out = new fake::Event();
    return hipSuccess;
}
inline hipError_t hipEventCreate(hipEvent_t *out) { return hipEventCreateWithFlags(out, 0); }
inline hipError_t hipEventDestroy(hipEvent_t e) {
    if (!e || !e->alive) {
        fake::report("hipEventDestroy of a dead event (destroyed twice?)", e);
        return hipErrorInvalidValue;
    }
    e->alive = false;
    delete e;  // a second destroy is then a heap-use-after-free for ASan as well
    return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t stream) {
    fake::Stream *s = fake::resolve(stream);
    e->stream = s;
    e->ticket = fake::enqueue(s);
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t e) {
    if (e->stream) fake::complete(e->stream, e->ticket);
    return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t event, unsigned) {
    fake::Stream *s = fake::resolve(stream);
    const uint64_t at = fake::enqueue(s);
    if (event && event->stream && event->stream != s) {
        std::lock_guard<std::mutex> lock(fake::state().mutex);
        s->waits.push_back({at, event->stream, event->ticket});
    }
    return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
    *ms = 0.f;
    return hipSuccess;
}
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

#define HIP_SYMBOL(x) (&(x))
template <typename T>
inline hipError_t hipMemcpyToSymbol(T *symbol, const void *src, size_t bytes) {
    memcpy((void *)symbol, src, std::min(bytes, sizeof(T)));
    return hipSuccess;
}
template <typename T>
inline hipError_t hipMemcpyFromSymbol(void *dst, T *symbol, size_t bytes) {
    memcpy(dst, (const void *)symbol, std::min(bytes, sizeof(T)));
    return hipSuccess;
}

namespace fake {
template <typename... Args>
inline void launch(hipStream_t stream, const Args &...args) {
    Stream *s = resolve(stream);
    const uint64_t ticket = enqueue(s);
    state().launches.fetch_add(1);
    (mark_arg(args, s, ticket), ...);
}
template <typename... Args>
inline void ext_launch(hipStream_t stream, hipEvent_t start, hipEvent_t stop, unsigned, const Args &...args) {
    Stream *s = resolve(stream);
    if (start) {
        start->stream = s;
        start->ticket = s->issued.load() + 1;
    }
    launch(stream, args...);
    if (stop) {
        stop->stream = s;
        stop->ticket = s->issued.load();
    }
}
}  // namespace fake
// the kernel itself is only ever named (stringified here, never evaluated): no device code exists in this build
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    ((void)(grid), (void)(block), (void)(lds), fake::launch(stream, __VA_ARGS__))
#define hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, start, stop, flags, ...) \
    ((void)(grid), (void)(block), (void)(lds), fake::ext_launch(stream, start, stop, flags, __VA_ARGS__))
