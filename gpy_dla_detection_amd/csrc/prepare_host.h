// The data half of NullGP.set_data (reference gpy_dla_detection/null_gp.py:117-177) and the bounds of the absorber redshifts
// (set_parameters.py:125-159), for one raw spectrum: pure host C++, no HIP and no device -- also built alone into the
// host-only sanitizer library (plan_host_debug.cpp) and driven by tests/test_host_logic.py against `null_gp.prepare_data`.
//
// Bit for bit what the Python does, operation by operation: every value is an IEEE sum, difference, product, quotient or
// comparison of doubles in the Python's order (this file is compiled without floating-point contraction), with two exceptions
// that go through a math library:
//   * `median ** 2` of a NumPy float64 scalar is libm's pow(median, 2.0) (scalarmath -> npy_pow), which is NOT always
//     median * median (0.09 % of arguments differ in the last bit on glibc 2.35): pow() here too, the same libm in the same process;
//   * the six padding pixels are 10 ** (log10(edge) +- k spacing) -- np.log10 and np.power, which NumPy dispatches to SVML on
//     AVX-512 hosts (4 % of results differ from libm's in the last bit).  The caller supplies them (`pad`): the Python binding
//     computes them with NumPy for the whole batch in one callback; a caller without NumPy takes `pad_libm`.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <string>
#include <vector>

// no fused multiply-adds in the functions that must round like the Python (x86-64's baseline has none anyway; this keeps
// a build with -march flags honest).  Scoped to those functions: this header is included in front of the device kernels.
#if defined(__clang__)
#define DLAGP_PREP_FN
#define DLAGP_PREP_BODY _Pragma("clang fp contract(off)")
#elif defined(__GNUC__)
#define DLAGP_PREP_FN __attribute__((optimize("fp-contract=off")))
#define DLAGP_PREP_BODY
#else
#define DLAGP_PREP_FN
#define DLAGP_PREP_BODY
#endif

namespace dlagp {
namespace prep {

// `Parameters` of the reference as the data half reads them (set_parameters.py:14-165); the *_cut are redshift differences
// (kms_to_z already applied, :105-110)
struct Params {
    double min_lambda, max_lambda;                                // modelled rest-frame window            (:33-34)
    double normalization_min_lambda, normalization_max_lambda;    // window of the flux normalisation      (:30-31)
    double pixel_spacing;                                         // log10 spacing of the padded grid      (:58)
    double lya_wavelength, lyman_limit;                           // Angstrom                              (:16-19)
    double min_z_cut, max_z_cut;
    int width;                                                    // padding pixels per side               (:57)
    int normalize;                                                // set_data(normalize=True)              (null_gp.py:121)
};

struct Spectrum {
    int num;                        // pixels
    const double *wavelengths;      // observed, Angstrom                 (read_spec's four arrays, run_bayes_select.py:147)
    const double *flux;
    const double *noise_variance;
    const uint8_t *pixel_mask;      // 1 = masked
    double z_qso;
};

struct Prepared {
    int n = 0, n_u = 0;
    double median = std::numeric_limits<double>::quiet_NaN();
    std::vector<double> x, y, v, this_wavelengths;   // [n] kept pixels: rest wavelength, normalised flux and noise variance, observed wavelength
    std::vector<double> unmasked_wavelengths;        // [n_u]
    std::vector<double> padded_wavelengths;          // [n_u + 2 width]; the 2 width padding entries are filled by `pad`
    std::vector<uint8_t> keep;                       // [n_u] = ~pixel_mask[ind_unmasked]    (dla_gp.py:360)
    double edge_lo = 0.0, edge_hi = 0.0;             // smallest / largest unmasked wavelength (np.min / np.max)
    double z_min = 0.0, z_max = 0.0;                 // min_z_dla / max_z_dla of this_wavelengths (the sample tables')
    double z_min_all = 0.0, z_max_all = 0.0;         // ... of the raw observed wavelengths (the driver's min_z_dlas / max_z_dlas)
};

// np.nanmedian of a 1-D float64 array as null_gp._nanmedian_1d does it: NaNs dropped, the middle element by partition, or
// the mean of the two middle ones; NaN for an empty window
DLAGP_PREP_FN inline double nanmedian(std::vector<double> &w) {
    DLAGP_PREP_BODY
    w.erase(std::remove_if(w.begin(), w.end(), [](double a) { return a != a; }), w.end());
    const size_t size = w.size();
    if (!size) return std::numeric_limits<double>::quiet_NaN();
    const size_t half = size / 2;
    std::nth_element(w.begin(), w.begin() + half, w.end());
    if (size % 2) return w[half];
    const double below = *std::max_element(w.begin(), w.begin() + half);  // the (half - 1)-th smallest
    return (below + w[half]) / 2.0;
}

// np.max([a, b]) / np.min([a, b]): NaN if either is
inline double np_max2(double a, double b) { return (a != a || b != b) ? std::numeric_limits<double>::quiet_NaN() : (a > b ? a : b); }
inline double np_min2(double a, double b) { return (a != a || b != b) ? std::numeric_limits<double>::quiet_NaN() : (a < b ? a : b); }

// (min_z_dla, max_z_dla) of `Parameters.z_dla_range` (set_parameters.py:125-159) over `count` observed wavelengths.
// np.min / np.max of the in-range ones (NaN-propagating); false when none is in range (np.min of an empty array raises).
DLAGP_PREP_FN inline bool z_dla_range(const Params &p, const double *wavelengths, int count, double z_qso, double *z_min, double *z_max) {
    DLAGP_PREP_BODY
    const double zp1 = 1.0 + z_qso;
    bool any = false, nan = false;
    double lo = 0.0, hi = 0.0;
    for (int j = 0; j < count; ++j) {
        const double rest = wavelengths[j] / zp1;  // emitted_wavelengths
        if (!(rest >= p.min_lambda && rest <= p.max_lambda)) continue;
        const double w = wavelengths[j];
        nan = nan || w != w;
        if (!any) lo = hi = w;
        else {
            lo = w < lo ? w : lo;
            hi = w > hi ? w : hi;
        }
        any = true;
    }
    if (!any) return false;
    if (nan) lo = hi = std::numeric_limits<double>::quiet_NaN();
    *z_min = np_max2(lo / p.lya_wavelength - 1, p.lyman_limit * zp1 / p.lya_wavelength - 1 + p.min_z_cut);
    *z_max = np_min2((hi / p.lya_wavelength - 1) - p.max_z_cut, z_qso - p.max_z_cut);
    return true;
}

// Everything of the data half except the values of the padding pixels.  Returns "" or what the Python would have raised.
DLAGP_PREP_FN inline std::string scan(const Params &p, const Spectrum &s, Prepared &out) {
    DLAGP_PREP_BODY
    if (s.num <= 0 || !s.wavelengths || !s.flux || !s.noise_variance || !s.pixel_mask) return "empty spectrum or NULL array";
    if (p.width < 1) return "width must be at least 1";
    const double zp1 = 1.0 + s.z_qso;
    std::vector<double> rest((size_t)s.num);
    for (int j = 0; j < s.num; ++j) rest[j] = s.wavelengths[j] / zp1;  // run_bayes_select.py:148, set_parameters.py:83-84
    double scale_y = 1.0, scale_v = 1.0;
    if (p.normalize) {  // null_gp.py:123-135
        std::vector<double> window;
        for (int j = 0; j < s.num; ++j)
            if (rest[j] >= p.normalization_min_lambda && rest[j] <= p.normalization_max_lambda && !s.pixel_mask[j])
                window.push_back(s.flux[j]);
        out.median = nanmedian(window);
        scale_y = out.median;
        // `median ** 2` of a NumPy float64 scalar: libm's pow, see the header -- through a volatile exponent, or the compiler
        // rewrites pow(x, 2.0) as x * x
        volatile double two = 2.0;
        scale_v = std::pow(out.median, two);
    }
    out.n = out.n_u = 0;
    for (int j = 0; j < s.num; ++j)
        if (rest[j] >= p.min_lambda && rest[j] <= p.max_lambda) {
            ++out.n_u;
            if (!s.pixel_mask[j]) ++out.n;
        }
    if (!out.n_u) return "no pixel inside the modelled window [min_lambda, max_lambda]";  // np.min of an empty array raises
    out.x.resize(out.n);
    out.y.resize(out.n);
    out.v.resize(out.n);
    out.this_wavelengths.resize(out.n);
    out.unmasked_wavelengths.resize(out.n_u);
    out.keep.resize(out.n_u);
    out.padded_wavelengths.assign((size_t)out.n_u + 2 * p.width, 0.0);
    bool nan = false;
    for (int j = 0, u = 0, k = 0; j < s.num; ++j) {
        if (!(rest[j] >= p.min_lambda && rest[j] <= p.max_lambda)) continue;
        const double observed = rest[j] * zp1;  // observed_wavelengths(x, z_qso): NOT the input wavelength to the last bit
        out.unmasked_wavelengths[u] = observed;
        out.padded_wavelengths[p.width + u] = observed;
        nan = nan || observed != observed;
        out.edge_lo = u == 0 ? observed : (observed < out.edge_lo ? observed : out.edge_lo);
        out.edge_hi = u == 0 ? observed : (observed > out.edge_hi ? observed : out.edge_hi);
        const bool kept = !s.pixel_mask[j];
        out.keep[u++] = kept ? 1 : 0;
        if (kept) {
            out.x[k] = rest[j];
            out.y[k] = p.normalize ? s.flux[j] / scale_y : s.flux[j];
            out.v[k] = p.normalize ? s.noise_variance[j] / scale_v : s.noise_variance[j];
            out.this_wavelengths[k] = observed;
            ++k;
        }
    }
    if (nan) out.edge_lo = out.edge_hi = std::numeric_limits<double>::quiet_NaN();
    // bounds of the absorber redshifts: of the kept pixels' observed wavelengths (the sample tables, dla_samples.py:94-104)
    // and of the raw spectrum (the driver's min_z_dlas / max_z_dlas, run_bayes_select.py:150-152)
    if (!z_dla_range(p, out.this_wavelengths.data(), out.n, s.z_qso, &out.z_min, &out.z_max))
        return "no kept pixel inside the modelled window";
    if (!z_dla_range(p, s.wavelengths, s.num, s.z_qso, &out.z_min_all, &out.z_max_all))
        return "no pixel inside the modelled window";
    return "";
}

// `pad[2 width]` -> the padding entries of padded_wavelengths: `width` below the grid, `width` above (null_gp.py:159-177)
inline void place_padding(const Params &p, const double *pad, Prepared &out) {
    for (int k = 0; k < p.width; ++k) {
        out.padded_wavelengths[k] = pad[k];
        out.padded_wavelengths[(size_t)p.width + out.n_u + k] = pad[p.width + k];
    }
}

// the padding pixels with libm: null_gp._logspace's arithmetic, 10 ** (start + i (stop - start) / (num - 1)), last point = stop
DLAGP_PREP_FN inline void pad_libm(const Params &p, double edge_lo, double edge_hi, double *pad) {
    DLAGP_PREP_BODY
    const double lo = std::log10(edge_lo), hi = std::log10(edge_hi);
    const double starts[2] = {lo - p.width * p.pixel_spacing, hi + p.pixel_spacing};
    const double stops[2] = {lo - p.pixel_spacing, hi + p.width * p.pixel_spacing};
    for (int side = 0; side < 2; ++side) {
        const double step = p.width > 1 ? (stops[side] - starts[side]) / (p.width - 1) : 0.0;
        for (int k = 0; k < p.width; ++k) {
            double e = (double)k * step;
            e += starts[side];
            if (k == p.width - 1) e = stops[side];
            pad[side * p.width + k] = std::pow(10.0, e);
        }
    }
}

// z_min + (z_max - z_min) * offsets  (samples.DLASamples._map_offsets; reference dla_samples.py:94-104): a product, then a sum
DLAGP_PREP_FN inline void sample_redshifts(double z_min, double z_max, const double *offsets, int64_t S, double *out) {
    DLAGP_PREP_BODY
    const double width = z_max - z_min;
    for (int64_t s = 0; s < S; ++s) {
        const double scaled = width * offsets[s];
        out[s] = z_min + scaled;
    }
}

}  // namespace prep
}  // namespace dlagp
