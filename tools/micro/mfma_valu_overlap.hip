// Does a v_mfma_f32_32x32x16_bf16 whose accumulator sits in ordinary registers ("v") keep the VALU
// of its SIMD from issuing, compared with an accumulator in accumulation registers ("a")?
// Each wave runs ITER x { 1 MFMA + F independent v_fma_f32 } ("2acc": two independent accumulation
// chains alternate) and reports cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <bool ACC_A, int F, int NACC>
__global__ void k(float* out, long long* cycles, int iters) {
    f32x16 acc, acc2;
    for (int i = 0; i < 16; ++i) acc[i] = acc2[i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * threadIdx.x); b[i] = (__bf16)1.0f; }
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 0.5f + i;
    const float c1 = 1.0001f, c2 = 0.5f;
    if constexpr (ACC_A) asm volatile("" : "+a"(acc), "+a"(acc2)); else asm volatile("" : "+v"(acc), "+v"(acc2));
    asm volatile("" : "+v"(a), "+v"(b));  // loop-invariant operands stay in their registers
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int it = 0; it < iters; ++it) {
        if constexpr (ACC_A)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
        else
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < F; ++j)
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j % 8]) : "v"(c1), "v"(c2));
        if constexpr (NACC == 2) {  // a second, independent accumulation chain
            if constexpr (ACC_A)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc2) : "v"(a), "v"(b));
            else
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < F; ++j)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j % 8]) : "v"(c1), "v"(c2));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0.f;
    if constexpr (ACC_A) asm volatile("" : "+a"(acc), "+a"(acc2)); else asm volatile("" : "+v"(acc), "+v"(acc2));
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <bool ACC_A, int F, int NACC = 1>
void run(int threads, const char* name) {
    const int blocks = 256, iters = 4000;
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks * (threads / 64));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<ACC_A, F, NACC><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e0);
    k<ACC_A, F, NACC><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    // s_memtime-style counter runs at 100 MHz on this chip; report wall ns per MFMA per wave as well
    printf("%-8s acc=%s F=%d waves/SIMD=%d: %.1f ns per MFMA per wave (kernel %.3f ms), counter ticks/MFMA %.2f\n", name,
           ACC_A ? "a" : "v", F, threads / 256, ms * 1e6 / (iters * NACC), ms, avg / (iters * NACC));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 512}) {
        run<false, 0>(threads, "mfma");
        run<true, 0>(threads, "mfma");
        run<false, 4>(threads, "mfma+4");
        run<true, 4>(threads, "mfma+4");
        run<false, 6>(threads, "mfma+6");
        run<true, 6>(threads, "mfma+6");
        run<false, 12>(threads, "mfma+12");
        run<true, 12>(threads, "mfma+12");
        run<false, 0, 2>(threads, "2acc");
        run<false, 4, 2>(threads, "2acc+4");
        run<false, 6, 2>(threads, "2acc+6");
        run<true, 6, 2>(threads, "2acc+6");
        run<false, 12, 2>(threads, "2acc+12");
    }
    return 0;
}
