// Host-only build of the launch planner (plan_host.h): no HIP, no device.
//
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared -fPIC plan_host_debug.cpp
//       -o libdlagp_plan_host.so            (`make sanitize`)
//
// Exports the same dlagp_plan_debug / dlagp_plan_cut as libdlagp_hip.so, so tests/test_planner_fuzz.py can drive the code
// that decides every index the accumulate kernel dereferences under AddressSanitizer + UBSan on the CPU.
#include <string>

#include "plan_host.h"

static thread_local std::string g_plan_error;

extern "C" {

const char *dlagp_last_error(void) { return g_plan_error.c_str(); }

int dlagp_plan_cut(int64_t groups, double chunks, int64_t forced_splits, int *full, double *frac) {
    if (!full || !frac || groups < 1 || !(chunks >= 1.0)) {
        g_plan_error = "bad argument";
        return DLAGP_EINVAL;
    }
    dlagp::plan::Options opt;
    opt.pixel_splits = forced_splits;
    dlagp::plan::choose_cut((long)groups, chunks, opt, full, frac);
    return DLAGP_OK;
}

int dlagp_plan_debug(int nb, const int *n_u, const uint8_t *const *keep, int ntaps, int64_t S, const int64_t *options,
                     int64_t *summary, int32_t *slot_pixel, int32_t *slot_pos, int64_t slots_cap, int32_t *passes,
                     int64_t passes_cap, int32_t *items, int64_t items_cap, uint8_t *counts, int64_t counts_cap) {
    return dlagp::plan::debug(nb, n_u, keep, ntaps, S, options, summary, slot_pixel, slot_pos, slots_cap, passes,
                              passes_cap, items, items_cap, counts, counts_cap, g_plan_error);
}

}  // extern "C"
