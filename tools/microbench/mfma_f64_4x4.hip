// v_mfma_f64_4x4x4_4b_f64 on gfx950: lane layout (discovered with one-hot operands) and issue cost, alone and
// mixed with v_mfma_f64_16x16x4_f64.  Motivation (accumulate_mfma): the rank-20 contraction needs 210
// w-columns + 20 q-columns; 16-column tiles waste 26 of 256.  If the 4-block 4x4x4 form is cheap enough, the two
// leftover groups of columns (2 of vech(M M'), 4 of M) can ride on it instead of on two full 64-cycle tiles.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_f64_4x4 mfma_f64_4x4.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

__global__ void small_kernel(const double *A, const double *B, double *D) {
    const int l = threadIdx.x;
    double d = 0.0;
    d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], d, 0, 0, 0);
    D[l] = d;
}

struct Stamp {
    unsigned long long cycles, realtime;
};

// NBIG 16x16x4 MFMAs and NSMALL 4x4x4 MFMAs per iteration, independent accumulators
template <int NBIG, int NSMALL>
__global__ __launch_bounds__(512) void mix_kernel(int iters, double seed, double *sink, Stamp *stamps) {
    const int l = threadIdx.x & 63;
    d4 acc[14];
    double sm[4] = {0, 0, 0, 0};
    for (int i = 0; i < 14; ++i) acc[i] = d4{0, 0, 0, 0};
    const double a = seed + l * 1e-3, b = 1.0 - l * 1e-4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NBIG; ++m) acc[m % 14] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m % 14], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NSMALL; ++m) sm[m & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, sm[m & 3], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = sm[0] + sm[1] + sm[2] + sm[3];
    for (int i = 0; i < 14; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps->cycles = t1 - t0;
        stamps->realtime = r1 - r0;
    }
}

template <int NBIG, int NSMALL>
static void run(const char *name, int waves_per_simd, double *sink, Stamp *stamps) {
    const int iters = 2000, threads = 256 * waves_per_simd, blocks = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mix_kernel<NBIG, NSMALL>), dim3(blocks), dim3(threads), 0, 0, 10, 1.25, sink, stamps);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix_kernel<NBIG, NSMALL>), dim3(blocks), dim3(threads), 0, 0, iters, 1.25, sink, stamps);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    Stamp st;
    CHECK(hipMemcpy(&st, stamps, sizeof(st), hipMemcpyDeviceToHost));
    printf("%-40s waves/SIMD=%d : %8.1f cycles/iter/wave  clock %.2f GHz  wall %.3f ms\n", name, waves_per_simd,
           (double)st.cycles / iters, (double)st.cycles / ((double)st.realtime * 10.0), ms);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main() {
    double *dA, *dB, *dD;
    CHECK(hipMalloc(&dA, 64 * 8));
    CHECK(hipMalloc(&dB, 64 * 8));
    CHECK(hipMalloc(&dD, 64 * 8));
    // one-hot A at lane la, B[lb] = lb + 1: D[lo] = sum of B values paired with A's lane for output lo
    std::vector<double> A(64), B(64), D(64);
    int pair_out[64][64];  // pair_out[la][lb] = output lane (or -1)
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) pair_out[la][lb] = -1;
    for (int lb = 0; lb < 64; ++lb) B[lb] = lb + 1;
    CHECK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    for (int la = 0; la < 64; ++la) {
        for (int i = 0; i < 64; ++i) A[i] = i == la ? 1.0 : 0.0;
        CHECK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(small_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        CHECK(hipMemcpy(D.data(), dD, 64 * 8, hipMemcpyDeviceToHost));
        for (int lo = 0; lo < 64; ++lo)
            if (D[lo] != 0.0) {
                const int lb = (int)D[lo] - 1;  // each (la, lo) pairs with exactly one lb if the product is a plain 4x4x4
                if (lb >= 0 && lb < 64 && D[lo] == (double)(lb + 1)) pair_out[la][lb] = lo;
                else printf("  la=%d lo=%d: D=%g is not a single B value\n", la, lo, D[lo]);
            }
    }
    // hypothesis: A lane = k*16 + blk*4 + i, B lane = k*16 + blk*4 + j, D lane = ?*; print what D lane (blk, i, j) lands in
    printf("pairs (la, lb) -> lo for la in the first block/k group:\n");
    for (int la = 0; la < 64; la += 1) {
        int cnt = 0;
        for (int lb = 0; lb < 64; ++lb) cnt += pair_out[la][lb] >= 0;
        if (la < 8 || la % 16 == 0) {
            printf("  la=%2d pairs with %d B lanes:", la, cnt);
            for (int lb = 0; lb < 64; ++lb)
                if (pair_out[la][lb] >= 0) printf(" (lb=%d -> lo=%d)", lb, pair_out[la][lb]);
            printf("\n");
        }
    }
    int bad = 0;
    for (int k = 0; k < 4; ++k)
        for (int blk = 0; blk < 4; ++blk)
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const int la = k * 16 + blk * 4 + i, lb = k * 16 + blk * 4 + j;
                    const int want = blk * 16 + i * 4 + j;  // candidate 1: D lane = blk*16 + i*4 + j
                    bad += pair_out[la][lb] != want;
                }
    printf("hypothesis A[k*16+blk*4+i] B[k*16+blk*4+j] -> D[blk*16+i*4+j]: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    bad = 0;
    for (int k = 0; k < 4; ++k)
        for (int blk = 0; blk < 4; ++blk)
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const int la = k * 16 + blk * 4 + i, lb = k * 16 + blk * 4 + j;
                    const int want = j * 16 + blk * 4 + i;  // candidate 2: D lane = j*16 + blk*4 + i (row-major like A)
                    bad += pair_out[la][lb] != want;
                }
    printf("hypothesis A[k*16+blk*4+i] B[k*16+blk*4+j] -> D[j*16+blk*4+i]: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);

    double *sink;
    Stamp *stamps;
    CHECK(hipMalloc(&sink, 256 * 8 * 256 * sizeof(double)));
    CHECK(hipMalloc(&stamps, sizeof(Stamp)));
    run<16, 0>("16 big", 1, sink, stamps);
    run<0, 16>("16 small", 1, sink, stamps);
    run<0, 16>("16 small", 2, sink, stamps);
    run<14, 2>("14 big + 2 small", 1, sink, stamps);
    run<14, 2>("14 big + 2 small", 2, sink, stamps);
    run<14, 0>("14 big", 1, sink, stamps);
    run<15, 1>("15 big + 1 small", 1, sink, stamps);
    return 0;
}
