// FP64 helpers for the hot kernels: reciprocal, exp of a non-positive argument, and a running
// product used to replace sum(log d) by one log per lane.  On gfx950 `log` costs ~400 cycles
// and an IEEE division ~55 per wave-instruction (profiles/r01_microbench_mfma_f64.txt), so the
// inner loops avoid both.
#pragma once
#include <hip/hip_runtime.h>

#include "faddeeva_table.h"

namespace dlagp {

// 1/x: v_rcp_f64 seed + two Newton steps (relative error ~1e-16 for normal x)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    return r;
}

// 1/x with a single Newton step: v_rcp_f64 is good to 4.6e-8, one step gives 2.2e-15
// (profiles/r01_microbench_dp_pipe.txt) -- enough for weights and wing terms
__device__ __forceinline__ double fast_rcp1(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(r, fma(-x, r, 1.0), r);
}

// 1/sqrt(x) for the Cholesky pivots: v_rsq_f64 seed + two coupled Newton (Goldschmidt) steps on g ~ sqrt(x) and
// h ~ 1/(2 sqrt(x)); relative error ~2e-16 for normal x.  Ten instructions instead of the nineteen of an IEEE sqrt
// followed by fast_rcp.  x <= 0, NaN and infinity give what v_rsq_f64 gives (NaN / infinity / 0): callers test the pivot.
__device__ __forceinline__ double fast_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    h = fma(h, r, h);
    return h + h;
}

// exp(x) for x <= 0 (also correct for moderate x > 0); max relative error 2e-14.
// x = k ln2 + r, |r| <= ln2/2, e^r by a degree-9 fit, scaled with v_ldexp_f64 (underflows to 0).
// One Horner step p * r + c with the coefficient forced into a scalar register pair.  hipcc
// otherwise parks the ten coefficients in 20 VGPRs and emits a v_mov_b64 + v_fmac_f64 pair per
// step; the VGPRs are needed for the MFMA accumulators and the copies cost issue slots.
__device__ __forceinline__ double horner_sgpr(double p, double r, double c) {
    double out;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(out) : "v"(p), "v"(r), "s"(c));
    return out;
}

__device__ __forceinline__ double fast_exp(double x) {
    constexpr double c[DLAGP_EXP_DEGREE + 1] = DLAGP_EXP_COEF_INIT;
    const double k = __builtin_rint(x * 1.44269504088896338700e+00);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = fma(c[DLAGP_EXP_DEGREE], r, c[DLAGP_EXP_DEGREE - 1]);
#pragma unroll
    for (int m = DLAGP_EXP_DEGREE - 2; m >= 0; --m) p = horner_sgpr(p, r, c[m]);
    // clamp the exponent so the int conversion is defined for -inf / huge arguments
    const double kc = fmax(k, -2200.0);
    return __builtin_amdgcn_ldexp(p, (int)kc);
}

// 2^t for t <= 0 (the kernel folds log2(e) into the column density, saving the range-reduction
// multiplies of fast_exp): k = rint(t), f = t - k exactly, 2^f by a degree-9 fit, v_ldexp_f64.
// v_cvt_i32_f64 saturates, so -inf / huge arguments give 2^INT_MIN = 0 and NaN stays NaN.
__device__ __forceinline__ double fast_exp2(double t) {
    constexpr double c[DLAGP_EXP_DEGREE + 1] = DLAGP_EXP2_COEF_INIT;
    const double k = __builtin_rint(t);
    const double f = t - k;
    double p = fma(c[DLAGP_EXP_DEGREE], f, c[DLAGP_EXP_DEGREE - 1]);
#pragma unroll
    for (int m = DLAGP_EXP_DEGREE - 2; m >= 0; --m) p = horner_sgpr(p, f, c[m]);
    int ki;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(ki) : "v"(k));
    return __builtin_amdgcn_ldexp(p, ki);
}

// The same with the second-highest coefficient handed in in a vector register (see vgpr_constant): the
// first Horner step is then one three-operand v_fma_f64 (scalar, vector, vector) instead of the
// v_mov_b64 + v_fmac_f64 pair hipcc emits for two literal coefficients.
__device__ __forceinline__ double fast_exp2(double t, double c8_vgpr) {
    constexpr double c[DLAGP_EXP_DEGREE + 1] = DLAGP_EXP2_COEF_INIT;
    const double k = __builtin_rint(t);
    const double f = t - k;
    double p;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "s"(c[DLAGP_EXP_DEGREE]), "v"(f), "v"(c8_vgpr));
#pragma unroll
    for (int m = DLAGP_EXP_DEGREE - 2; m >= 0; --m) p = horner_sgpr(p, f, c[m]);
    int ki;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(ki) : "v"(k));
    return __builtin_amdgcn_ldexp(p, ki);
}

// 2^t with the coefficients read from `c` (constant memory -> SGPRs), `c8_vgpr` = c[8] in a vector register
__device__ __forceinline__ double fast_exp2_table(double t, const double *__restrict__ c, double c8_vgpr) {
    const double k = __builtin_rint(t);
    const double f = t - k;
    double p = fma(c[9], f, c8_vgpr);
#pragma unroll
    for (int m = 7; m >= 0; --m) p = fma(p, f, c[m]);
    int ki;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(ki) : "v"(k));  // saturating: -inf / huge arguments give 2^INT_MIN = 0
    return __builtin_amdgcn_ldexp(p, ki);
}

// a constant pinned in a vector register (opaque to the optimiser, so it is not folded back into a literal)
__device__ __forceinline__ double vgpr_constant(double c) {
    asm volatile("" : "+v"(c));
    return c;
}

// Running product with explicit exponent: value = mant * 2^expo.  sum(log d_i) = log(mant) + expo ln2.
// 750 multiplications lose < 1e-13 absolutely in the logarithm (against |log p| ~ 1e3).
struct LogProduct {
    double mant = 1.0;
    int expo = 0;
    __device__ __forceinline__ void multiply(double d) {
        const double t = mant * d;
        mant = __builtin_amdgcn_frexp_mant(t);
        expo += __builtin_amdgcn_frexp_exp(t);
    }
    __device__ __forceinline__ double log_value() const {
        return log(mant) + expo * 6.93147180559945286227e-01;
    }
};

// broadcast one lane's double to the whole wave through SGPRs (two v_readlane_b32); `src_lane`
// must be wave-uniform
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

}  // namespace dlagp

