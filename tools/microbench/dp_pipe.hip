// Does FP64 VALU work execute concurrently with FP64 MFMA of a sibling wave on the same SIMD (gfx950)?
// 512-thread workgroups = 2 waves per SIMD: waves 0-3 run role A, waves 4-7 role B.
//   role 0: idle   role 1: v_mfma_f64_16x16x4 stream   role 2: 16 independent v_fma_f64 chains
//   role 3: 16 independent v_fma_f32 chains   role 4: integer v_mad_u32 chains
// If the pipes were independent, time(A|B) = max(time(A), time(B)); if shared, the sum.
// Also: accuracy of the raw v_rcp_f64 seed and after one / two Newton steps.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ROLE>
__device__ __forceinline__ double run_role(int iters, double seed, int lane) {
    if (ROLE == 1) {
        d4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
        double a = seed + lane * 1e-3, b = 1.0 - lane * 1e-4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m & 7], 0, 0, 0);
        }
        double s = 0;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        return s;
    } else if (ROLE == 2) {
        double v[16];
        for (int i = 0; i < 16; ++i) v[i] = seed + i + lane * 1e-2;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 256; ++q) v[q & 15] = __builtin_fma(v[q & 15], 0.999999, 1e-7);
        }
        double s = 0;
        for (int i = 0; i < 16; ++i) s += v[i];
        return s;
    } else if (ROLE == 3) {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = (float)seed + i + lane * 1e-2f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 256; ++q) v[q & 15] = __builtin_fmaf(v[q & 15], 0.9999f, 1e-3f);
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) s += v[i];
        return s;
    } else if (ROLE == 4) {
        unsigned v[16];
        for (int i = 0; i < 16; ++i) v[i] = (unsigned)seed + i + lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 256; ++q) v[q & 15] = v[q & 15] * 1664525u + 1013904223u;
        }
        unsigned s = 0;
        for (int i = 0; i < 16; ++i) s += v[i];
        return (double)s;
    }
    return 0.0;
}

template <int A, int B>
__global__ __launch_bounds__(512) void pair_kernel(int iters, double seed, double *sink) {
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    double s = wave < 4 ? run_role<A>(iters, seed, lane) : run_role<B>(iters, seed, lane);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int A, int B>
static float time_pair(double *sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((pair_kernel<A, B>), dim3(256), dim3(512), 0, 0, 10, 1.25, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((pair_kernel<A, B>), dim3(256), dim3(512), 0, 0, 2000, 1.25, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    return best;
}

__global__ void rcp_kernel(const double *x, double *r0, double *r1, double *r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    double e = fma(-v, r, 1.0);
    r = fma(r, e, r);
    r1[i] = r;
    e = fma(-v, r, 1.0);
    r = fma(r, e, r);
    r2[i] = r;
}

int main() {
    double *sink;
    CHECK(hipMalloc(&sink, 256 * 512 * sizeof(double)));
    const char *names[] = {"idle", "mfma_f64", "fma_f64", "fma_f32", "mad_u32"};
    float t10 = time_pair<1, 0>(sink), t20 = time_pair<2, 0>(sink), t30 = time_pair<3, 0>(sink), t40 = time_pair<4, 0>(sink);
    printf("alone (waves 0-3 active, 4-7 idle): mfma_f64 %.3f ms, fma_f64 %.3f ms, fma_f32 %.3f ms, mad_u32 %.3f ms\n", t10, t20, t30, t40);
    printf("mfma_f64 | mfma_f64 : %.3f ms (sum %.3f, max %.3f)\n", time_pair<1, 1>(sink), 2 * t10, t10);
    printf("mfma_f64 | fma_f64  : %.3f ms (sum %.3f, max %.3f)\n", time_pair<1, 2>(sink), t10 + t20, fmaxf(t10, t20));
    printf("mfma_f64 | fma_f32  : %.3f ms (sum %.3f, max %.3f)\n", time_pair<1, 3>(sink), t10 + t30, fmaxf(t10, t30));
    printf("mfma_f64 | mad_u32  : %.3f ms (sum %.3f, max %.3f)\n", time_pair<1, 4>(sink), t10 + t40, fmaxf(t10, t40));
    printf("fma_f64  | fma_f64  : %.3f ms (sum %.3f, max %.3f)\n", time_pair<2, 2>(sink), 2 * t20, t20);
    printf("fma_f64  | fma_f32  : %.3f ms (sum %.3f, max %.3f)\n", time_pair<2, 3>(sink), t20 + t30, fmaxf(t20, t30));
    (void)names;

    const int n = 1 << 20;
    std::vector<double> hx(n), h0(n), h1(n), h2(n);
    srand(1);
    for (int i = 0; i < n; ++i) hx[i] = ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 200) - 100);
    double *dx, *d0, *d1, *d2;
    CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&d0, n * 8)); CHECK(hipMalloc(&d1, n * 8)); CHECK(hipMalloc(&d2, n * 8));
    CHECK(hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rcp_kernel, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    CHECK(hipMemcpy(h0.data(), d0, n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h1.data(), d1, n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h2.data(), d2, n * 8, hipMemcpyDeviceToHost));
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) {
        long double ex = 1.0L / (long double)hx[i];
        m0 = fmax(m0, fabs((double)((h0[i] - ex) / ex)));
        m1 = fmax(m1, fabs((double)((h1[i] - ex) / ex)));
        m2 = fmax(m2, fabs((double)((h2[i] - ex) / ex)));
    }
    printf("v_rcp_f64 max relative error: seed %.3e, +1 Newton %.3e, +2 Newton %.3e\n", m0, m1, m2);
    return 0;
}
