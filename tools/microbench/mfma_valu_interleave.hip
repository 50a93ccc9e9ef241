// Does vector work cost less when it is interleaved between a wave's own MFMAs than when it runs in a
// separate phase?  (gfx950, FP64).  256 workgroups x 512 threads = two waves per SIMD, all doing the same:
//   phases      : per iteration 16 x v_mfma_f64_16x16x4, then NV independent v_fma_f64
//   interleaved : per iteration 16 x (v_mfma_f64_16x16x4, NV/16 v_fma_f64)
//   hipcc -O3 --offload-arch=gfx950 -o mfma_valu_interleave mfma_valu_interleave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NV, bool INTERLEAVED>
__global__ __launch_bounds__(512) void kernel(int iters, double seed, double *sink) {
    d4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x * 1e-3;
    const double a = 0.999999, b = 1e-7;
    constexpr int PER = NV / 16;
    for (int it = 0; it < iters; ++it) {
        if (INTERLEAVED) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < PER; ++q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[(m * PER + q) & 15]) : "v"(a), "v"(b));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[q & 15]) : "v"(a), "v"(b));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + acc[i][0];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, bool I>
static float run(double *sink, int threads) {
    const int iters = 400;
    hipLaunchKernelGGL((kernel<NV, I>), dim3(256), dim3(threads), 0, 0, 20, 1.25, sink);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((kernel<NV, I>), dim3(256), dim3(threads), 0, 0, iters, 1.25, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

template <int NV>
static void both(double *sink, float base1, float base2) {
    for (int threads : {256, 512}) {
        const float p = run<NV, false>(sink, threads), i = run<NV, true>(sink, threads);
        const float base = threads == 256 ? base1 : base2;
        printf("%d waves/SIMD, %3d FMAs per 16 MFMAs: phases %.3f ms (+%.3f over MFMA only), interleaved %.3f ms (+%.3f)\n",
               threads / 256, NV, p, p - base, i, i - base);
    }
}

int main() {
    double *sink;
    CHECK(hipMalloc(&sink, 256 * 512 * sizeof(double)));
    for (int i = 0; i < 60; ++i) hipLaunchKernelGGL((kernel<0, false>), dim3(256), dim3(512), 0, 0, 400, 1.25, sink);
    CHECK(hipDeviceSynchronize());
    const float b1 = run<0, false>(sink, 256), b2 = run<0, false>(sink, 512);
    printf("MFMA only: %.3f ms (1 wave/SIMD), %.3f ms (2 waves/SIMD)\n", b1, b2);
    both<16>(sink, b1, b2);
    both<64>(sink, b1, b2);
    both<128>(sink, b1, b2);
    both<256>(sink, b1, b2);
    return 0;
}
