// Shared host/device helpers for the gfx950 kernels of this library.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/nanovllm_hip.h"

namespace nv {

constexpr int kWave = 64;       // CDNA wavefront width
constexpr int kHeadDim = 128;   // the only head_dim the kernels are specialised for

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// ---- error reporting -------------------------------------------------------------
void set_error(const char* fmt, ...);
nv_tuning tuning();   // snapshot of the process-wide launch-plan overrides (nv_set_tuning)

#define NV_REQUIRE(cond, code, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            ::nv::set_error(__VA_ARGS__);    \
            return (code);                   \
        }                                    \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return NV_ERR_LAUNCH;
    }
    return NV_OK;
}

// ---- division by a launch constant -------------------------------------------------
// n / d for 0 <= n < 2^31 and 1 <= d < 2^31 as one 32 x 32 -> 64-bit multiply and a shift (the host picks
// shr = 31 + ceil(log2 d) and mul = ceil(2^shr / d) < 2^32; the product's error n * (mul * d - 2^shr) stays
// below 2^shr, so the quotient is exact).  A kernel that decomposes its block index by launch constants pays
// about 25 instructions per division otherwise, which is most of a short kernel's instruction count.
struct FastDiv {
    uint32_t mul, shr;
};
inline FastDiv make_fastdiv(uint32_t d) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    FastDiv f;
    f.shr = 31 + l;
    f.mul = (uint32_t)(((1ull << f.shr) + d - 1) / d);
    return f;
}
__device__ __forceinline__ uint32_t fast_div(uint32_t n, FastDiv f) { return (uint32_t)(((uint64_t)n * f.mul) >> f.shr); }

// ---- bf16 <-> f32 bit helpers ----------------------------------------------------
// A uint32 holds two bf16: element 0 in the low half, element 1 in the high half.
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// Round-to-nearest-even f32 -> bf16 through the compiler's native conversion
// (v_cvt_pk_bf16_f32 on gfx950; keeps NaN a NaN).
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    bf16x2 p;
    p[0] = (__bf16)lo;
    p[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ float round_bf16(float x) { return (float)(__bf16)x; }

// 16-byte streaming load that is read once: non-temporal hint keeps it from displacing
// reusable lines in L2 / Infinity Cache.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ uint4 load_stream16(const uint4* p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// ---- cross-lane helpers (wave64) --------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// Sum over each aligned group of 16 lanes (a DPP "row"); every lane gets the total.
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_f32<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_f32<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_f32<0x141>(x);  // row_half_mirror
    x += dpp_f32<0x140>(x);  // row_mirror
    return x;
}
__device__ __forceinline__ float wave_sum(float x) {
    x = row16_sum(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

}  // namespace nv
