// TEST INFRASTRUCTURE (host-only sanitizer build, see fake_device.h): the few kernel NAMES and device symbols that the host
// side of dlagp_hip.hip mentions outside a launch macro -- function addresses handed to hipFuncSetAttribute, a kernel
// chosen into a variable, the constant-memory symbols it uploads.  Bodies are empty: no device code exists in this build.
#pragma once
#include "host_shared.h"

namespace dlagp {
inline LineConstants DLAGP_LINES;
inline unsigned long long DLAGP_STAMPS[8];
template <bool ROW16, bool NATIVE>
void finalize_rank20(const double *, int, long, long, const SightlineDev *, double *, int *, const unsigned char *) {}
template <bool CDF_IN_LDS>
void level_finish(const double *, long, int, int, SampleSource, double, const double *, double *, double *, int *, double *, int *,
                  const int *, int *) {}
inline void accumulate_generic() {}
inline void absorption_generic() {}
}  // namespace dlagp
