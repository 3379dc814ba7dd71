// How many single-issue instructions fit, for free, between two v_mfma_f32_32x32x16_bf16 of ONE wave on a
// SIMD?  Each kernel is one hand-written block: UNROLL x { MFMA ; F fillers } with the accumulators named
// explicitly (NCH independent chains, in accumulation registers "a" or ordinary registers "v"), repeated
// in a loop, timed with s_memtime.  Reports shader cycles per MFMA at one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_gap_fillers mfma_gap_fillers.hip && ./mfma_gap_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define STR2(x) #x
#define STR(x) STR2(x)

// fillers write v[100..111] (independent of each other and of the MFMAs), read v[112..116]
#define FMA(i) "v_fma_f32 v" STR(i) ", v112, v113, v" STR(i) "\n\t"
#define EXP(i) "v_exp_f32 v" STR(i) ", v114\n\t"
#define CVT(i) "v_cvt_pk_bf16_f32 v" STR(i) ", v115, v116\n\t"
#define MAX3(i) "v_max3_f32 v" STR(i) ", v112, v113, v" STR(i) "\n\t"

#define F0 ""
#define F2 FMA(100) FMA(101)
#define F3 FMA(100) FMA(101) FMA(102)
#define F4 FMA(100) FMA(101) FMA(102) FMA(103)
#define F5 FMA(100) FMA(101) FMA(102) FMA(103) FMA(104)
#define F6 FMA(100) FMA(101) FMA(102) FMA(103) FMA(104) FMA(105)
#define F8 F4 FMA(104) FMA(105) FMA(106) FMA(107)
#define FE4 EXP(108) FMA(100) FMA(101) FMA(102) FMA(103)      /* 1 exp + 4 fma */
#define FE2C2 EXP(108) FMA(100) FMA(101) CVT(109) MAX3(110)   /* softmax-like mix */
#define FEE3 EXP(108) EXP(111) FMA(100) FMA(101) FMA(102)      /* 2 exp + 3 */

#define MFMA_A(c) "v_mfma_f32_32x32x16_bf16 a[" STR(c) ":" STR(c) "+15], v[120:123], v[124:127], a[" STR(c) ":" STR(c) "+15]\n\t"
#define MFMA_V(c) "v_mfma_f32_32x32x16_bf16 v[" STR(c) ":" STR(c) "+15], v[120:123], v[124:127], v[" STR(c) ":" STR(c) "+15]\n\t"

#define BODY4(M, F) M(0) F M(16) F M(32) F M(48) F
#define BODY1(M, F) M(0) F M(0) F M(0) F M(0) F
#define BODY8(M, F) M(0) F M(16) F M(32) F M(48) F M(64) F M(80) F M(96) F M(112) F

#define KERNEL(name, BODY)                                                                      \
    __global__ __launch_bounds__(512) void name(long long* cycles, int iters) {                \
        long long t0, t1;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));                       \
        for (int it = 0; it < iters; ++it) {                                                     \
            asm volatile(BODY BODY BODY BODY ::: "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", \
                         "v108", "v109", "v110", "v111", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", \
                         "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24",  \
                         "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38",  \
                         "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52",  \
                         "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v112", "v113", "v114",\
                         "v115", "v116", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127",                     \
                         "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
                         "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29",   \
                         "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43",   \
                         "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57",   \
                         "a58", "a59", "a60", "a61", "a62", "a63", "a127");                                                   \
        }                                                                                        \
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));  \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    }

KERNEL(a4_f0, BODY4(MFMA_A, F0))
KERNEL(a4_f2, BODY4(MFMA_A, F2))
KERNEL(a4_f3, BODY4(MFMA_A, F3))
KERNEL(a4_f4, BODY4(MFMA_A, F4))
KERNEL(a4_f5, BODY4(MFMA_A, F5))
KERNEL(a4_f6, BODY4(MFMA_A, F6))
KERNEL(a4_f8, BODY4(MFMA_A, F8))
KERNEL(a4_fe4, BODY4(MFMA_A, FE4))
KERNEL(a4_fe2c2, BODY4(MFMA_A, FE2C2))
KERNEL(a4_fee3, BODY4(MFMA_A, FEE3))
KERNEL(v4_f0, BODY4(MFMA_V, F0))
KERNEL(v4_f4, BODY4(MFMA_V, F4))
KERNEL(v4_f5, BODY4(MFMA_V, F5))
KERNEL(v4_fe4, BODY4(MFMA_V, FE4))
KERNEL(a1_f0, BODY1(MFMA_A, F0))
KERNEL(a1_f4, BODY1(MFMA_A, F4))
KERNEL(a1_f5, BODY1(MFMA_A, F5))
KERNEL(v1_f4, BODY1(MFMA_V, F4))

typedef void (*kern_t)(long long*, int);

static void run(kern_t k, const char* name, int threads) {
    const int blocks = 256, iters = 500, per_iter = 16;
    long long* cyc;
    hipMalloc(&cyc, sizeof(long long) * blocks * (threads / 64));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, cyc, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= h.size();
    printf("%-10s waves/SIMD=%d: %6.2f cycles per MFMA per wave, %6.2f ns (=> %.2f GHz)\n", name, threads / 256,
           avg / (iters * per_iter), ms * 1e6 / (iters * per_iter), avg / (ms * 1e6));
    hipFree(cyc);
}

int main() {
#define R(k) run(k, #k, threads)
    for (int threads : {256, 512}) {
        R(a4_f0); R(a4_f2); R(a4_f3); R(a4_f4); R(a4_f5); R(a4_f6); R(a4_f8); R(a4_fe4); R(a4_fe2c2); R(a4_fee3);
        R(v4_f0); R(v4_f4); R(v4_f5); R(v4_fe4); R(a1_f0); R(a1_f4); R(a1_f5); R(v1_f4);
    }
    return 0;
}
