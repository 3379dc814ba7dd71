// The per-head arithmetic of the q/k-norm + rotary step, shared by the kernel that runs it as a launch of
// its own (norm_rope.hip) and by the decode attention kernel that runs it in front of its key stream
// (attn_decode.hip): one function, so both produce the same bits.
// A head (128 bf16) is handled by 8 lanes: lane c of the aligned group of 8 holds x1[8c..8c+7] and
// x2[8c..8c+7] (x1 = first half, x2 = second half of the head).
#pragma once

#include "nv_common.h"

namespace nv {

__device__ __forceinline__ float group8_sum(float x) {
    x += dpp_f32<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_f32<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_f32<0x141>(x);  // row_half_mirror: the other quad of the aligned 8 lanes
    return x;
}

template <bool NORM>
__device__ __forceinline__ void head_norm_rope(const uint4 a, const uint4 b,      // the head's x1 / x2 pieces
                                               const uint4 wa, const uint4 wb,    // norm weights (NORM only)
                                               const float4 c0, const float4 c1,  // cos[8c .. 8c+7]
                                               const float4 s0, const float4 s1,  // sin[8c .. 8c+7]
                                               float eps, uint4& out1, uint4& out2) {
    const uint32_t au[4] = {a.x, a.y, a.z, a.w};
    const uint32_t bu[4] = {b.x, b.y, b.z, b.w};
    float x1[8], x2[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x1[2 * j] = bf16_lo(au[j]);
        x1[2 * j + 1] = bf16_hi(au[j]);
        x2[2 * j] = bf16_lo(bu[j]);
        x2[2 * j + 1] = bf16_hi(bu[j]);
    }
    if (NORM) {
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += x1[j] * x1[j] + x2[j] * x2[j];
        ss = group8_sum(ss);
        const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
        const uint32_t wau[4] = {wa.x, wa.y, wa.z, wa.w};
        const uint32_t wbu[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // the norm module's bf16 output is what the rotary module reads
            x1[2 * j] = round_bf16(x1[2 * j] * r * bf16_lo(wau[j]));
            x1[2 * j + 1] = round_bf16(x1[2 * j + 1] * r * bf16_hi(wau[j]));
            x2[2 * j] = round_bf16(x2[2 * j] * r * bf16_lo(wbu[j]));
            x2[2 * j + 1] = round_bf16(x2[2 * j + 1] * r * bf16_hi(wbu[j]));
        }
    }
    const float co[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float si[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    uint32_t o1[4], o2[4];
    {
        // The reference evaluates x1*cos - x2*sin as separate fp32 tensor ops: every product
        // is rounded before the add.  Keep the compiler from fusing them into FMAs.
#pragma clang fp contract(off)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a1 = x1[2 * j] * co[2 * j], a2 = x2[2 * j] * si[2 * j];
            const float b1 = x1[2 * j + 1] * co[2 * j + 1], b2 = x2[2 * j + 1] * si[2 * j + 1];
            const float c1 = x1[2 * j] * si[2 * j], c2 = x2[2 * j] * co[2 * j];
            const float d1 = x1[2 * j + 1] * si[2 * j + 1], d2 = x2[2 * j + 1] * co[2 * j + 1];
            o1[j] = pack_bf16(a1 - a2, b1 - b2);
            o2[j] = pack_bf16(c1 + c2, d1 + d2);
        }
    }
    out1 = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    out2 = make_uint4(o2[0], o2[1], o2[2], o2[3]);
}

}  // namespace nv
