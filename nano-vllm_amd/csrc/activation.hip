// SiluAndMul: y[r, j] = silu(x[r, j]) * x[r, I + j] for rows of 2*I bf16 (the fused
// gate/up projection output).  Elementwise and HBM-bound: 16-byte loads and stores, fp32
// arithmetic, one rounding to bf16, as the compiled reference evaluates
// nanovllm/layers/activation.py:10-16.
#include "nv_common.h"

namespace nv {

__device__ __forceinline__ float silu_mul(float g, float u) {
    // g * sigmoid(g) * u; expf(-g) overflows to +inf for g < -88 and the quotient becomes -0
    return g / (1.0f + __expf(-g)) * u;
}

__global__ __launch_bounds__(256) void silu_and_mul_kernel(const uint4* __restrict__ x, uint4* __restrict__ y,
                                                           int64_t rows, int inter_vec,
                                                           int64_t x_stride_vec) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * inter_vec) return;
    const int64_t row = idx / inter_vec;
    const int col = (int)(idx - row * inter_vec);
    const uint4 g = x[row * x_stride_vec + col];
    const uint4 u = x[row * x_stride_vec + inter_vec + col];
    uint4 o;
    o.x = pack_bf16(silu_mul(bf16_lo(g.x), bf16_lo(u.x)), silu_mul(bf16_hi(g.x), bf16_hi(u.x)));
    o.y = pack_bf16(silu_mul(bf16_lo(g.y), bf16_lo(u.y)), silu_mul(bf16_hi(g.y), bf16_hi(u.y)));
    o.z = pack_bf16(silu_mul(bf16_lo(g.z), bf16_lo(u.z)), silu_mul(bf16_hi(g.z), bf16_hi(u.z)));
    o.w = pack_bf16(silu_mul(bf16_lo(g.w), bf16_lo(u.w)), silu_mul(bf16_hi(g.w), bf16_hi(u.w)));
    y[idx] = o;
}

}  // namespace nv

extern "C" int nv_silu_and_mul(const void* x, void* y, int64_t rows, int inter, int64_t x_stride,
                               void* stream) {
    using namespace nv;
    NV_REQUIRE(rows >= 0 && inter > 0, NV_ERR_INVALID_ARG, "nv_silu_and_mul: bad sizes rows=%lld inter=%d",
               (long long)rows, inter);
    if (rows == 0) return NV_OK;
    NV_REQUIRE(x && y, NV_ERR_INVALID_ARG, "nv_silu_and_mul: null pointer");
    NV_REQUIRE(inter % 8 == 0 && x_stride % 8 == 0 && x_stride >= 2 * (int64_t)inter, NV_ERR_UNSUPPORTED,
               "nv_silu_and_mul: inter=%d stride=%lld (need inter%%8==0, stride%%8==0, stride>=2*inter)",
               inter, (long long)x_stride);
    NV_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0, NV_ERR_INVALID_ARG,
               "nv_silu_and_mul: pointers must be 16-byte aligned");
    const int64_t n = rows * (inter / 8);
    const int64_t grid = (n + 255) / 256;
    NV_REQUIRE(grid < (1LL << 31), NV_ERR_UNSUPPORTED, "nv_silu_and_mul: launch too large");
    hipLaunchKernelGGL(silu_and_mul_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)x, (uint4*)y, rows, inter / 8, x_stride / 8);
    return check_launch("nv_silu_and_mul");
}
