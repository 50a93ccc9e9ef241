// Micro-benchmarks that the kernel design in DESIGN.md rests on (gfx950):
//   1. lane layout of v_mfma_f64_16x16x4_f64 (A, B, C/D), checked with exact integer data
//   2. issue interval of that MFMA per SIMD, at 1 and 2 waves per SIMD
//   3. FP64 VALU rates (fma, rcp, exp, log) per SIMD
//   4. how much FP64 VALU work hides under the MFMAs of the SAME wave / of a sibling wave
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_f64 mfma_f64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e = (x);                                                             \
        if (e != hipSuccess) {                                                          \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__);        \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

// ---- 1. layout --------------------------------------------------------------------------------
__global__ void layout_kernel(const double *A /*16x4*/, const double *B /*4x16*/, double *C /*16x16*/) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];   // A[row = l&15][k = l>>4]
    const double b = B[(l >> 4) * 16 + (l & 15)];  // B[k = l>>4][col = l&15]
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];  // row = (l>>4) + 4 r
}

// ---- timing kernels ---------------------------------------------------------------------------
struct Stamp {
    unsigned long long cycles, realtime;
};

template <int NMFMA, int NVALU, int VKIND>
__global__ __launch_bounds__(512) void mix_kernel(int iters, double seed, double *sink, Stamp *stamps, int role_split) {
    const int l = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    d4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = seed + l * 1e-3, b = 1.0 - l * 1e-4;
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + l * 1e-2;
    // role_split: 0 = every wave does both; 1 = waves 0-3 MFMA only, waves 4-7 (the second wave of
    // each SIMD in a 512-thread workgroup) VALU only
    const bool do_mfma = role_split == 0 || wave < 4;
    const bool do_valu = role_split == 0 || wave >= 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int m = 0; m < NMFMA; ++m) acc[m & 15] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m & 15], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int q = 0; q < NVALU; ++q) {
                double &x = v[q & 7];
                if (VKIND == 0) x = __builtin_fma(x, 0.999999, 1e-7);
                if (VKIND == 1) x = __builtin_amdgcn_rcp(x) + 1.5;
                if (VKIND == 2) x = exp(-x * 1e-3) + 1.0;
                if (VKIND == 3) x = log(x) + 2.0;
                if (VKIND == 4) x = 1.0 / x + 1.5;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps->cycles = t1 - t0;
        stamps->realtime = r1 - r0;
    }
}

template <int NMFMA, int NVALU, int VKIND>
static void run(const char *name, int waves_per_simd, int role_split, double *sink, Stamp *stamps) {
    const int iters = 2000;
    const int threads = 256 * waves_per_simd;  // 4 waves per SIMD-wave: one workgroup per CU
    const int blocks = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mix_kernel<NMFMA, NVALU, VKIND>), dim3(blocks), dim3(threads), 0, 0, 10, 1.25, sink, stamps, role_split);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix_kernel<NMFMA, NVALU, VKIND>), dim3(blocks), dim3(threads), 0, 0, iters, 1.25, sink, stamps, role_split);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    Stamp st;
    CHECK(hipMemcpy(&st, stamps, sizeof(st), hipMemcpyDeviceToHost));
    const double clock_ghz = (double)st.cycles / ((double)st.realtime * 10.0);  // realtime ticks at 100 MHz
    const double cyc_per_iter = (double)st.cycles / iters;
    const double tflops = NMFMA ? (double)blocks * (role_split ? 4 : 4 * waves_per_simd) * iters * NMFMA * 2048.0 / (ms * 1e-3) / 1e12 : 0.0;
    printf("%-44s waves/SIMD=%d split=%d : %8.1f cycles/iter/wave  clock %.2f GHz  wall %.3f ms  mfma %.1f TFLOP/s\n", name,
           waves_per_simd, role_split, cyc_per_iter, clock_ghz, ms, tflops);
}

int main() {
    // layout check with exact integers and asymmetric operands
    std::vector<double> A(64), B(64), C(256), ref(256, 0.0);
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 3;
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + k * 11 + j * j;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
            for (int k = 0; k < 4; ++k) ref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dC;
    CHECK(hipMalloc(&dA, 64 * 8));
    CHECK(hipMalloc(&dB, 64 * 8));
    CHECK(hipMalloc(&dC, 256 * 8));
    CHECK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    CHECK(hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += C[i] != ref[i];
    printf("layout v_mfma_f64_16x16x4_f64: A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4r col=l&15 : %s (%d mismatches)\n",
           bad ? "WRONG" : "OK", bad);

    double *sink;
    Stamp *stamps;
    CHECK(hipMalloc(&sink, 256 * 8 * 256 * sizeof(double)));
    CHECK(hipMalloc(&stamps, sizeof(Stamp)));

    run<16, 0, 0>("16 mfma", 1, 0, sink, stamps);
    run<16, 0, 0>("16 mfma", 2, 0, sink, stamps);
    run<0, 64, 0>("64 fma", 1, 0, sink, stamps);
    run<0, 64, 0>("64 fma", 2, 0, sink, stamps);
    run<0, 64, 1>("64 rcp(+add)", 1, 0, sink, stamps);
    run<0, 64, 4>("64 div(+add)", 1, 0, sink, stamps);
    run<0, 16, 2>("16 exp(+mul,add)", 1, 0, sink, stamps);
    run<0, 16, 3>("16 log(+add)", 1, 0, sink, stamps);
    run<16, 64, 0>("16 mfma + 64 fma same wave", 1, 0, sink, stamps);
    run<16, 128, 0>("16 mfma + 128 fma same wave", 1, 0, sink, stamps);
    run<16, 256, 0>("16 mfma + 256 fma same wave", 1, 0, sink, stamps);
    run<16, 128, 0>("16 mfma + 128 fma same wave", 2, 0, sink, stamps);
    run<16, 256, 0>("16 mfma + 256 fma same wave", 2, 0, sink, stamps);
    run<16, 256, 0>("16 mfma | 256 fma sibling waves", 2, 1, sink, stamps);
    run<16, 128, 0>("16 mfma | 128 fma sibling waves", 2, 1, sink, stamps);
    return 0;
}
