// Host-only build of the launch planner (plan_host.h): no HIP, no device.
//
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared -fPIC plan_host_debug.cpp
//       -o libdlagp_plan_host.so            (`make sanitize`)
//
// Exports the same dlagp_plan_debug / dlagp_plan_cut / dlagp_prepare_spectrum as libdlagp_hip.so, so tests/test_planner_fuzz.py can drive the code
// that decides every index the accumulate kernel dereferences under AddressSanitizer + UBSan on the CPU.
#include <string>

#include "plan_host.h"
#include "prepare_host.h"

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

// the data half of NullGP.set_data for one spectrum (prepare_host.h), as libdlagp_hip.so exports it
int dlagp_prepare_spectrum(const dlagp_prepare_params *params, int num_pixels, const double *wavelengths, const double *flux,
                           const double *noise_variance, const uint8_t *pixel_mask, double z_qso, const double *pad, int *n,
                           int *n_u, double *scalars, double *x, double *y, double *v, double *this_wavelengths,
                           double *unmasked_wavelengths, double *padded_wavelengths, uint8_t *keep) {
    namespace prep = dlagp::prep;
    if (!params) {
        g_plan_error = "params is NULL";
        return DLAGP_EINVAL;
    }
    const prep::Params p{params->min_lambda, params->max_lambda, params->normalization_min_lambda,
                         params->normalization_max_lambda, params->pixel_spacing, params->lya_wavelength, params->lyman_limit,
                         params->min_z_cut, params->max_z_cut, params->width, params->normalize};
    prep::Prepared out;
    g_plan_error = prep::scan(p, prep::Spectrum{num_pixels, wavelengths, flux, noise_variance, pixel_mask, z_qso}, out);
    if (!g_plan_error.empty()) return DLAGP_EINVAL;
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
}

}  // extern "C"
