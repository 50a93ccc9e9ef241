// Device Voigt / Faddeeva evaluation for the Lyman-series absorption profile (FP64).
//
// Replaces scipy.special.wofz at the reference's call site gpy_dla_detection/voigt.py:247-248.
// Only Re w(x + i y) is needed, and y = gamma_l / (sqrt2 sigma) takes 31 fixed, tiny values
// (7e-8 .. 4.8e-4), so the function is evaluated as a Taylor series in y about the real axis:
//
//   Re w(x+iy) = e^{-x^2} [1 - y^2 (2x^2-1) + y^4 (16x^4-48x^2+12)/24]
//                - (2y/sqrt(pi)) [G(x) - (y^2/6) ((4x^2-6) G(x) + 2)] + O(y^5),
//   G(x) = 1 - 2 x Dawson(x).
//
//   |x| <  16 : G from a degree-14 Taylor table (faddeeva_table.h), e^{-x^2} evaluated
//   |x| >= 16 : e^{-x^2} < 1e-111 is dropped and G is replaced by its asymptotic series,
//               giving  Re w = (y/sqrt(pi)) u [A(u) - y^2 u B(u)],  u = 1/x^2,
//               A(u) = sum_m (2m+1)!!/2^m u^m (10 terms),  B(u) = 1 + 5u + 26.25u^2
//
// tests/test_gpu_parity.py::test_device_voigt_profile_matches_reference_golden holds both forms the kernels use
// (dlagp_debug_voigt) to the reference's own Voigt values on golden grid G1, all 31 lines, and records the
// measured maximum relative differences (profiles/r02_faddeeva_accuracy.txt).
#pragma once
#include <hip/hip_runtime.h>

#include "dlagp_device.h"
#include "faddeeva_table.h"

__constant__ double DLAGP_FAD_TABLE[DLAGP_FAD_INTERVALS * (DLAGP_FAD_DEGREE + 1)] = DLAGP_FAD_TABLE_INIT;
__constant__ LineConstants DLAGP_LINES;

namespace dlagp {

// u [A(u) - y^2 u B(u)] for |x| >= 16; multiply by y/sqrt(pi) for Re w.
__device__ __forceinline__ double wing_series(double x, double y2) {
    const double u = 1.0 / (x * x);
    double a = 1278767.724609375;  // (2m+1)!!/2^m, m = 9 .. 0
    a = fma(a, u, 134607.12890625);
    a = fma(a, u, 15836.1328125);
    a = fma(a, u, 2111.484375);
    a = fma(a, u, 324.84375);
    a = fma(a, u, 59.0625);
    a = fma(a, u, 13.125);
    a = fma(a, u, 3.75);
    a = fma(a, u, 1.5);
    a = fma(a, u, 1.0);
    const double b = fma(fma(26.25, u, 5.0), u, 1.0);
    return u * fma(-y2 * u, b, a);
}

// Re w(x + i y) for |x| < 16.
__device__ __forceinline__ double core_taylor(double x, double y) {
    const double ax = fabs(x);
    int idx = (int)(ax * DLAGP_FAD_INV_WIDTH);
    idx = idx > DLAGP_FAD_INTERVALS - 1 ? DLAGP_FAD_INTERVALS - 1 : idx;
    const double t = ax - (idx + 0.5) * DLAGP_FAD_WIDTH;
    const double *g = DLAGP_FAD_TABLE + idx * (DLAGP_FAD_DEGREE + 1);
    double G = g[DLAGP_FAD_DEGREE];
#pragma unroll
    for (int m = DLAGP_FAD_DEGREE - 1; m >= 0; --m) G = fma(G, t, g[m]);
    const double x2 = ax * ax;
    const double y2 = y * y;
    const double gauss = exp(-x2);
    const double pe = 1.0 - y2 * (2.0 * x2 - 1.0) + y2 * y2 * (16.0 * x2 * x2 - 48.0 * x2 + 12.0) * (1.0 / 24.0);
    const double pg = G - (y2 * (1.0 / 6.0)) * ((4.0 * x2 - 6.0) * G + 2.0);
    return gauss * pe - (2.0 * 0.564189583547756286948 /* 1/sqrt(pi) */) * y * pg;
}

// Optical depth per unit column density of `num_lines` series members at one wavelength:
//   sum_l lc_l Voigt(v_l; sigma, gamma_l)   (voigt.py:304-309, without the sign and the N factor).
// `mult[l]` = c / (lambda_l (1 + z)) / 1e8 (voigt.py:299).
__device__ __forceinline__ double line_optical_depth(double lam, const double *mult, int num_lines) {
    double tau = 0.0;
    for (int l = 0; l < num_lines; ++l) {
        // voigt.py:305: velocity = wavelengths * multipliers[l] - c (product rounded first)
        const double velocity = __dsub_rn(__dmul_rn(lam, mult[l]), DLAGP_LINES.c);
        const double x = velocity * DLAGP_LINES.inv_sqrt2_sigma;
        double term;
        if (fabs(x) >= DLAGP_FAD_XSPLIT) {
            term = DLAGP_LINES.k_wing[l] * wing_series(x, DLAGP_LINES.y[l] * DLAGP_LINES.y[l]);
        } else {
            term = DLAGP_LINES.k_core[l] * core_taylor(x, DLAGP_LINES.y[l]);
        }
        if (term == term) tau += term;  // np.nansum over the lines (voigt.py:309)
    }
    return tau;
}

// Multipliers of one absorber (voigt.py:299), evaluated exactly as the reference does.
__device__ __forceinline__ double line_multiplier(int l, double z) {
    return DLAGP_LINES.c / (DLAGP_LINES.lambda_cm[l] * (1.0 + z)) / 1e8;
}

}  // namespace dlagp
