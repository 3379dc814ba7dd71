// What this GPU sustains on bf16 MFMAs alone: one wave per SIMD (4 waves per workgroup, as the hand-scheduled
// prefill kernel runs), each issuing independent v_mfma_f32_32x32x16_bf16 back to back on register operands.
// bench.py launches it next to the prefill records: the clock a chip holds under matrix load differs from box
// to box, and this number is the ceiling that any MFMA kernel on THIS box is priced against besides the nominal
// peak.  Measurement tool only (loaded by bench.py through ctypes); not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libmfma_ceiling.so mfma_ceiling.hip
#include <hip/hip_runtime.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// random_operands = 0: every operand element is (nearly) the same value -- few bits toggle, the chip holds its
// highest clock; 1: operands are pseudo-random values in (-2, 2) that differ per lane, element and MFMA, as
// attention data does -- the datapath toggles, power rises and the clock the chip holds is lower.
__device__ inline float pseudo_random(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(int)(x & 0xffff) * (1.0f / 16384.0f) - 2.0f;
}

template <bool RANDOM>
__global__ __launch_bounds__(256) void mfma_ceiling_kernel(float* sink, int iters, float seed) {
    bf16x8 a, b, a2, b2;
    for (int i = 0; i < 8; ++i) {
        const unsigned id = (blockIdx.x * 256 + threadIdx.x) * 32 + i;
        a[i] = (__bf16)(RANDOM ? pseudo_random(id) : seed + (float)(threadIdx.x & 7) * 0.001f);
        b[i] = (__bf16)(RANDOM ? pseudo_random(id + 8) : seed - (float)(threadIdx.x & 3) * 0.001f);
        a2[i] = RANDOM ? (__bf16)pseudo_random(id + 16) : a[i];
        b2[i] = RANDOM ? (__bf16)pseudo_random(id + 24) : b[i];
    }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c7, 0, 0, 0);
        }
    }
    f32x16 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += s[i];
    if (t == 12345.678f) sink[blockIdx.x * 256 + threadIdx.x] = t;      // keeps the chains alive; never true in practice
}

// The other bf16 shape, 16x16x32 (same flops per cycle: 16 cycles per instruction), random operands.
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256) void mfma_ceiling_16x16_kernel(float* sink, int iters) {
    bf16x8 a, b, a2, b2;
    for (int i = 0; i < 8; ++i) {
        const unsigned id = (blockIdx.x * 256 + threadIdx.x) * 32 + i;
        a[i] = (__bf16)pseudo_random(id);
        b[i] = (__bf16)pseudo_random(id + 8);
        a2[i] = (__bf16)pseudo_random(id + 16);
        b2[i] = (__bf16)pseudo_random(id + 24);
    }
    f32x4 c[16] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((k & 1) ? a2 : a, (k & 2) ? b2 : b, c[k], 0, 0, 0);
        }
    }
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += c[k][0] + c[k][1] + c[k][2] + c[k][3];
    if (t == 12345.678f) sink[blockIdx.x * 256 + threadIdx.x] = t;
}

extern "C" {
// MFMAs per wave per launch = 32 * iters; flops per MFMA = 2 * 32 * 32 * 16.
__attribute__((visibility("default"))) double mfma_ceiling_flops(int blocks, int iters) {
    return (double)blocks * 4.0 * 32.0 * (double)iters * 2.0 * 32.0 * 32.0 * 16.0;
}
// sink: device buffer of blocks * 256 floats (never written in practice).  Returns the hipError_t of the launch.
__attribute__((visibility("default"))) int mfma_ceiling_launch(void* sink, int blocks, int iters, int random_operands, void* stream) {
    if (sink == nullptr || blocks <= 0 || iters <= 0) return -1;
    if (random_operands == 2)     // 64 MFMAs of half the flops per iteration: the same flops per launch
        hipLaunchKernelGGL(mfma_ceiling_16x16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)sink, iters);
    else if (random_operands)
        hipLaunchKernelGGL(mfma_ceiling_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)sink, iters, 1.0f);
    else
        hipLaunchKernelGGL(mfma_ceiling_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)sink, iters, 1.0f);
    return (int)hipGetLastError();
}
}
