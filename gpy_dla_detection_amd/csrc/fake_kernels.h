// TEST INFRASTRUCTURE (host-only sanitizer build, see fake_device.h): the few kernel NAMES and device symbols that the host
// side of dlagp_hip.hip mentions outside a launch macro -- function addresses handed to hipFuncSetAttribute, a kernel
// chosen into a variable, the constant-memory symbols it uploads.  Bodies are empty: no device code exists in this build.
#pragma once
#include "host_shared.h"

namespace dlagp {
inline LineConstants DLAGP_LINES;
inline unsigned long long DLAGP_STAMPS[8];
template <bool ROW16, bool NATIVE>
void finalize_rank20(const double *, int, long, long, const SightlineDev *, double *, int *, const unsigned char *, double * = nullptr,
                     long = 0, EvidenceTail = EvidenceTail{}) {}  // the real kernel's parameter list (kernels_common.h): the host side takes its address
template <bool CUM_IN_LDS>
void level_finish_fused(LevelArgs) {}
inline void level_weights(LevelArgs) {}
inline void level_resample(LevelArgs) {}
inline void level_weights_two(LevelArgsTwo) {}
inline void level_resample_two(LevelArgsTwo) {}
inline void accumulate_generic() {}
inline void absorption_generic() {}
}  // namespace dlagp

// What a launch touches through a struct that travels BY VALUE (fake::launch marks pointer arguments itself).  In namespace
// fake: the structs are global, so argument-dependent lookup finds these overloads through the fake::Stream argument only
// (round 6: they sat in namespace dlagp and were never called).
namespace fake {
// what a segment copy touches: the device array of every piece
inline void mark_arg(const CopyTable &t, Stream *s, uint64_t ticket) {
    for (int i = 0; i < t.count; ++i) {
        // a piece registered before its array was allocated (round 6: the null model's flag, downloaded by a call whose
        // side launch -- the allocation -- had moved behind the registration: a GPU fault on the real device)
        if (!t.seg[i].device_ptr && t.seg[i].bytes) report("copy_segments piece with a NULL device address", nullptr);
        mark(reinterpret_cast<const void *>(t.seg[i].device_ptr), s, ticket);
    }
}
// what a level-finish launch touches (launch marks pointer arguments; this struct travels by value)
inline void mark_arg(const LevelArgs &a, Stream *s, uint64_t t) {
    for (const void *p : {(const void *)a.ll, (const void *)a.uniforms, (const void *)a.sample_ll, (const void *)a.evidences,
                          (const void *)a.base, (const void *)a.weights, (const void *)a.exch, (const void *)a.alive,
                          (const void *)a.status, (const void *)a.ties})
        mark(p, s, t);
}
inline void mark_arg(const EvidenceTail &e, Stream *s, uint64_t t) {
    for (const void *p : {(const void *)e.evidence, (const void *)e.too_close, (const void *)e.part, (const void *)e.tickets}) mark(p, s, t);
}
inline void mark_arg(const LevelArgsTwo &p, Stream *s, uint64_t t) {
    mark_arg(p.first, s, t);
    mark_arg(p.second, s, t);
}
}  // namespace fake
