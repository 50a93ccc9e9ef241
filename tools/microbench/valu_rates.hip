// Issue cost of the FP64 / integer vector instructions the accumulate kernel uses (gfx950), in cycles per
// wave-instruction: 256 workgroups x 512 threads = two waves per SIMD, each running 16 independent chains
// of one instruction.  Shader clock from s_memtime around the loop (wave 0 of each workgroup).
//   hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define OP_LIST(X) \
    X(0, "v_fma_f64", "v_fma_f64 %0, %0, %2, %3") \
    X(1, "v_mul_f64", "v_mul_f64 %0, %0, %2") \
    X(2, "v_add_f64", "v_add_f64 %0, %0, %3") \
    X(3, "v_max_f64", "v_max_f64 %0, %0, %3") \
    X(4, "v_rcp_f64", "v_rcp_f64 %0, %0") \
    X(5, "v_rndne_f64", "v_rndne_f64 %0, %0") \
    X(6, "v_ldexp_f64", "v_ldexp_f64 %0, %0, 1") \
    X(7, "v_frexp_mant_f64", "v_frexp_mant_f64 %0, %0") \
    X(8, "v_mov_b64", "v_mov_b64 %0, %2") \
    X(9, "v_cmp_lt_f64", "v_cmp_lt_f64 vcc, %0, %2") \
    X(10, "v_cvt_i32_f64", "v_cvt_i32_f64 %1, %0") \
    X(11, "v_lshl_add_u32", "v_lshl_add_u32 %1, %1, 1, %1") \
    X(12, "v_mov_b32", "v_mov_b32 %1, %1") \
    X(13, "v_add_u32", "v_add_u32 %1, %1, %1") \
    X(14, "v_sqrt_f64", "v_sqrt_f64 %0, %0") \
    X(15, "v_mfma_f64_16x16x4", "")

template <int OP>
__global__ __launch_bounds__(512) void rate_kernel(int iters, double seed, double *sink, unsigned long long *cycles) {
    double v[16];
    int w[16];
    const double a = 0.999999, b = 1e-7;
    for (int i = 0; i < 16; ++i) {
        v[i] = seed + i + threadIdx.x * 1e-3;
        w[i] = threadIdx.x + i;
    }
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 256; ++q) {
            if (OP == 15) {
                acc[q & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q & 7], 0, 0, 0);
            } else {
#define X(ID, NAME, TEXT) if (OP == ID) asm volatile(TEXT : "+v"(v[q & 15]), "+v"(w[q & 15]) : "v"(a), "v"(b) : "vcc");
                OP_LIST(X)
#undef X
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + w[i];
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    // every wave reports its own start and end: with two waves per SIMD the older wave can run its MFMAs back to back
    // while the other one waits, so one wave's elapsed time says nothing about the SIMD's time per instruction
    if ((threadIdx.x & 63) == 0) {
        cycles[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t0;
        cycles[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = t1;
    }
}

template <int OP>
static void run(const char *name, double *sink, unsigned long long *cyc, int threads) {
    const int iters = 200;
    hipLaunchKernelGGL((rate_kernel<OP>), dim3(256), dim3(threads), 0, 0, 20, 1.25, sink, cyc);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate_kernel<OP>), dim3(256), dim3(threads), 0, 0, iters, 1.25, sink, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long h[256 * 8 * 2];
    CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    const int waves = threads / 64;
    double mean_wave = 0, mean_span = 0;
    for (int b = 0; b < 256; ++b) {
        unsigned long long first = ~0ull, last = 0;
        for (int w = 0; w < waves; ++w) {
            const unsigned long long t0 = h[(b * 8 + w) * 2], t1 = h[(b * 8 + w) * 2 + 1];
            mean_wave += (double)(t1 - t0) / (256.0 * waves);
            first = t0 < first ? t0 : first;
            last = t1 > last ? t1 : last;
        }
        mean_span += (double)(last - first) / 256.0;
    }
    const double waves_per_simd = threads / 256.0;  // 4 SIMDs per CU
    // SIMD time per instruction: the workgroup's span (first start to last end) over the instructions one SIMD issued
    printf("%-28s %d waves/SIMD: %.3f ms, %.2f clk elapsed per instruction of a wave (mean over waves), %.2f clk of SIMD time each\n",
           name, (int)waves_per_simd, ms, mean_wave / (iters * 256.0), mean_span / (iters * 256.0 * waves_per_simd));
}

int main() {
    double *sink;
    unsigned long long *cyc;
    CHECK(hipMalloc(&sink, 256 * 512 * sizeof(double)));
    CHECK(hipMalloc(&cyc, 256 * 8 * 2 * sizeof(unsigned long long)));
    // warm the clocks
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((rate_kernel<0>), dim3(256), dim3(512), 0, 0, 200, 1.25, sink, cyc);
    CHECK(hipDeviceSynchronize());
    for (int threads : {256, 512}) {
#define X(ID, NAME, TEXT) run<ID>(NAME, sink, cyc, threads);
        OP_LIST(X)
#undef X
    }
    return 0;
}
