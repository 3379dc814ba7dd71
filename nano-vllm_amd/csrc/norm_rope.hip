// RMSNorm, residual-add + RMSNorm, rotary embedding, and the fused per-head
// q/k-norm + rotary + KV-cache scatter.  All HBM-bound elementwise/row-reduction work:
// 16-byte vector loads/stores, fp32 arithmetic, DPP reductions inside a wave.
//
// Semantics follow nanovllm/layers/layernorm.py:79-100 (single rounding to bf16 after the
// weight multiply, as the compiled reference produces) and nanovllm/layers/rope.py:5-56.
#include "nv_common.h"
#include "norm_rope_shared.h"

namespace nv {

// ------------------------------------------------------------------------------------
// RMSNorm over rows of `cols` bf16.  One wave per row (WAVE_PER_ROW) for short rows,
// one 256-thread workgroup per row otherwise.  Each lane keeps its share of the row
// (<= MAXV 16-byte vectors) in registers between the reduction and the scaling pass.
// ------------------------------------------------------------------------------------
template <bool ADD, bool WAVE_PER_ROW, int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const uint4* __restrict__ x,
                                                      const uint4* __restrict__ res,
                                                      const uint4* __restrict__ w,
                                                      uint4* __restrict__ y,
                                                      uint4* __restrict__ new_res, int64_t rows,
                                                      int cols_vec, int64_t x_stride_vec, float eps,
                                                      float inv_cols) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int64_t row;
    int first, step;
    if (WAVE_PER_ROW) {
        row = (int64_t)blockIdx.x * 4 + wave;
        first = lane;
        step = 64;
    } else {
        row = blockIdx.x;
        first = threadIdx.x;
        step = 256;
    }
    const bool active = row < rows;
    const uint4* xr = x + (active ? row : 0) * x_stride_vec;
    const uint4* rr = ADD ? res + (active ? row : 0) * (int64_t)cols_vec : nullptr;

    // Every load of the row is issued before the first use: x, the residual and the weights arrive in one
    // round trip (at decode sizes the kernel is a chain of memory latencies, not a stream).
    float f[MAXV][8];
    uint4 wv[MAXV], xv[MAXV], rv[MAXV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = first + i * step;
        if (active && c < cols_vec) {
            xv[i] = xr[c];
            if (ADD) rv[i] = rr[c];
            wv[i] = w[c];
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = first + i * step;
        if (active && c < cols_vec) {
            const uint32_t xu[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
            if (ADD) {
                const uint32_t ru[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
                uint32_t nu[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f[i][2 * j] = bf16_lo(xu[j]) + bf16_lo(ru[j]);
                    f[i][2 * j + 1] = bf16_hi(xu[j]) + bf16_hi(ru[j]);
                    nu[j] = pack_bf16(f[i][2 * j], f[i][2 * j + 1]);
                }
                new_res[row * cols_vec + c] = make_uint4(nu[0], nu[1], nu[2], nu[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f[i][2 * j] = bf16_lo(xu[j]);
                    f[i][2 * j + 1] = bf16_hi(xu[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += f[i][j] * f[i][j];
        }
    }
    ss = wave_sum(ss);
    if (!WAVE_PER_ROW) {
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        ss = red[0] + red[1] + red[2] + red[3];
    }
    const float r = rsqrtf(ss * inv_cols + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = first + i * step;
        if (active && c < cols_vec) {
            const uint32_t wu[4] = {wv[i].x, wv[i].y, wv[i].z, wv[i].w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = pack_bf16(f[i][2 * j] * r * bf16_lo(wu[j]),
                                 f[i][2 * j + 1] * r * bf16_hi(wu[j]));
            }
            y[row * cols_vec + c] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

template <bool ADD>
static int launch_rmsnorm(const void* x, const void* res, const void* w, void* y, void* new_res,
                          int64_t rows, int cols, int64_t x_stride, float eps, hipStream_t st) {
    const int cols_vec = cols / 8;
    const float inv = 1.0f / (float)cols;
    const int64_t xs = x_stride / 8;
#define NV_RMS_LAUNCH(WPR, MAXV, GRID)                                                          \
    hipLaunchKernelGGL((rmsnorm_kernel<ADD, WPR, MAXV>), dim3((unsigned)(GRID)), dim3(256), 0, st, \
                       (const uint4*)x, (const uint4*)res, (const uint4*)w, (uint4*)y,          \
                       (uint4*)new_res, rows, cols_vec, xs, eps, inv)
    if (cols_vec <= 64) NV_RMS_LAUNCH(true, 1, (rows + 3) / 4);
    else if (cols_vec <= 256) NV_RMS_LAUNCH(true, 4, (rows + 3) / 4);
    else if (cols_vec <= 1024) NV_RMS_LAUNCH(false, 4, rows);
    else NV_RMS_LAUNCH(false, 8, rows);
#undef NV_RMS_LAUNCH
    return check_launch(ADD ? "nv_add_rmsnorm" : "nv_rmsnorm");
}

// ------------------------------------------------------------------------------------
// Rotary embedding and the fused q/k-norm + rotary + cache scatter.
// One workgroup per token.  A head (128 bf16) is handled by 8 lanes: lane c of the group
// holds x1[8c..8c+7] and x2[8c..8c+7] (x1 = first half, x2 = second half of the head).
// ------------------------------------------------------------------------------------
// At decode sizes (a few hundred tokens) this kernel is a chain of memory latencies, so everything that
// does not depend on the token's position is fetched first and in one go -- the first 32 heads, both norm
// weights, the token's V row -- and only the cos / sin row waits for positions[t]; later passes (more than
// 32 heads) fetch their heads one pass ahead.
template <bool NORM>
__global__ __launch_bounds__(256) void qknorm_rope_store_kernel(
    const uint4* __restrict__ q, const uint4* __restrict__ k, const uint4* __restrict__ v,
    const uint4* __restrict__ q_w, const uint4* __restrict__ k_w, const float4* __restrict__ cos_sin,
    const int64_t* __restrict__ positions, const int32_t* __restrict__ slot_mapping,
    uint4* __restrict__ k_cache, uint4* __restrict__ v_cache, uint4* __restrict__ q_out,
    uint4* __restrict__ k_out, int H_q, int H_kv, int64_t max_pos, float eps, int64_t num_slots,
    int64_t q_stride_vec, int64_t k_stride_vec, int64_t v_stride_vec) {
    const int t = blockIdx.x;
    const int c = threadIdx.x & 7;
    const int group = threadIdx.x >> 3;  // 32 head-groups per pass
    const int n_heads = H_q + H_kv;
    const uint4* qt = q + (int64_t)t * q_stride_vec;
    const uint4* kt = k + (int64_t)t * k_stride_vec;
    // Every lane of a wave runs the same trip count so the DPP reductions see valid
    // neighbours; lanes whose head is out of range work on head 0 and discard.
    auto head_src = [&](int hh) {
        const int h = hh < n_heads ? hh : 0;
        return h < H_q ? qt + (int64_t)h * 16 : kt + (int64_t)(h - H_q) * 16;
    };
    const uint4* src = head_src(group);
    uint4 a = src[c], b = src[8 + c];
    const uint4 zero = make_uint4(0, 0, 0, 0);
    uint4 qwa = zero, qwb = zero, kwa = zero, kwb = zero;
    if (NORM) {
        qwa = q_w[c];
        qwb = q_w[8 + c];
        kwa = k_w[c];
        kwb = k_w[8 + c];
    }
    const int v_vecs = H_kv * 16;
    uint4 v0 = zero;
    if (v && (int)threadIdx.x < v_vecs) v0 = v[(int64_t)t * v_stride_vec + threadIdx.x];

    int64_t pos = positions[t];
    pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
    const float4* cs = cos_sin + pos * 32;  // 128 fp32 per position
    const float4 c0 = cs[2 * c], c1 = cs[2 * c + 1];            // cos[8c .. 8c+7]
    const float4 s0 = cs[16 + 2 * c], s1 = cs[16 + 2 * c + 1];  // sin[8c .. 8c+7]
    int64_t slot = -1;
    if (slot_mapping) {
        slot = slot_mapping[t];
        if (slot >= num_slots) slot = -1;
    }
    if (slot >= 0 && v) {
        uint4* vd = v_cache + slot * H_kv * 16;
        if ((int)threadIdx.x < v_vecs) vd[threadIdx.x] = v0;
        const uint4* vs = v + (int64_t)t * v_stride_vec;
        for (int i = threadIdx.x + blockDim.x; i < v_vecs; i += blockDim.x) vd[i] = vs[i];   // more than 16 kv heads
    }

    const int passes = (n_heads + 31) / 32;
    for (int p = 0; p < passes; ++p) {
        const int hh = p * 32 + group;
        const bool valid = hh < n_heads;
        const int h = valid ? hh : 0;
        const bool is_q = h < H_q;
        const uint4 a_cur = a, b_cur = b;
        if (p + 1 < passes) {
            const uint4* nsrc = head_src(hh + 32);
            a = nsrc[c];
            b = nsrc[8 + c];
        }
        uint4 o1, o2;
        head_norm_rope<NORM>(a_cur, b_cur, is_q ? qwa : kwa, is_q ? qwb : kwb, c0, c1, s0, s1, eps, o1, o2);
        if (!valid) continue;
        if (is_q) {
            uint4* dst = q_out + ((int64_t)t * H_q + h) * 16;
            dst[c] = o1;
            dst[8 + c] = o2;
        } else {
            const int hk = h - H_q;
            if (k_out) {
                uint4* dst = k_out + ((int64_t)t * H_kv + hk) * 16;
                dst[c] = o1;
                dst[8 + c] = o2;
            }
            if (slot >= 0) {
                uint4* dst = k_cache + (slot * H_kv + hk) * 16;
                dst[c] = o1;
                dst[8 + c] = o2;
            }
        }
    }
}

}  // namespace nv

// ---------------------------------- C ABI ------------------------------------------
extern "C" int nv_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int cols,
                          int64_t x_stride, float eps, void* stream) {
    using namespace nv;
    NV_REQUIRE(rows >= 0 && cols > 0, NV_ERR_INVALID_ARG, "nv_rmsnorm: bad sizes rows=%lld cols=%d",
               (long long)rows, cols);
    if (rows == 0) return NV_OK;
    NV_REQUIRE(x && w && y, NV_ERR_INVALID_ARG, "nv_rmsnorm: null pointer");
    NV_REQUIRE(cols % 8 == 0 && cols <= 16384 && x_stride % 8 == 0 && x_stride >= cols,
               NV_ERR_UNSUPPORTED, "nv_rmsnorm: cols=%d stride=%lld (need cols%%8==0, <=16384)", cols,
               (long long)x_stride);
    NV_REQUIRE(((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) % 16 == 0, NV_ERR_INVALID_ARG,
               "nv_rmsnorm: pointers must be 16-byte aligned");
    return launch_rmsnorm<false>(x, nullptr, w, y, nullptr, rows, cols, x_stride, eps,
                                 (hipStream_t)stream);
}

extern "C" int nv_add_rmsnorm(const void* x, const void* res, const void* w, void* y, void* new_res,
                              int64_t rows, int cols, float eps, void* stream) {
    using namespace nv;
    NV_REQUIRE(rows >= 0 && cols > 0, NV_ERR_INVALID_ARG,
               "nv_add_rmsnorm: bad sizes rows=%lld cols=%d", (long long)rows, cols);
    if (rows == 0) return NV_OK;
    NV_REQUIRE(x && res && w && y && new_res, NV_ERR_INVALID_ARG, "nv_add_rmsnorm: null pointer");
    NV_REQUIRE(cols % 8 == 0 && cols <= 16384, NV_ERR_UNSUPPORTED,
               "nv_add_rmsnorm: cols=%d (need cols%%8==0, <=16384)", cols);
    NV_REQUIRE(((uintptr_t)x | (uintptr_t)res | (uintptr_t)w | (uintptr_t)y | (uintptr_t)new_res) % 16 == 0,
               NV_ERR_INVALID_ARG, "nv_add_rmsnorm: pointers must be 16-byte aligned");
    return launch_rmsnorm<true>(x, res, w, y, new_res, rows, cols, cols, eps, (hipStream_t)stream);
}

static int qknorm_rope_store_impl(const char* name, const void* q, const void* k, const void* v,
                                  const void* q_w, const void* k_w, const float* cos_sin,
                                  const int64_t* positions, const int32_t* slot_mapping,
                                  void* k_cache, void* v_cache, void* q_out, void* k_out, int T,
                                  int H_q, int H_kv, int D, int64_t max_pos, float eps,
                                  int64_t num_blocks, int block_size, int64_t q_stride_t,
                                  int64_t k_stride_t, int64_t v_stride_t, void* stream) {
    using namespace nv;
    NV_REQUIRE(T >= 0 && H_q > 0 && H_kv > 0 && max_pos > 0, NV_ERR_INVALID_ARG,
               "%s: bad sizes T=%d H_q=%d H_kv=%d max_pos=%lld", name, T, H_q, H_kv,
               (long long)max_pos);
    NV_REQUIRE(D == kHeadDim, NV_ERR_UNSUPPORTED, "%s: head_dim %d unsupported (128 only)", name, D);
    if (T == 0) return NV_OK;
    NV_REQUIRE(q && k && cos_sin && positions && q_out, NV_ERR_INVALID_ARG, "%s: null pointer", name);
    NV_REQUIRE((q_w == nullptr) == (k_w == nullptr), NV_ERR_INVALID_ARG,
               "%s: q_w and k_w must both be given or both be NULL", name);
    const bool store = slot_mapping != nullptr;
    if (store) {
        NV_REQUIRE(k_cache && v_cache && v && num_blocks > 0 && block_size > 0, NV_ERR_INVALID_ARG,
                   "%s: scatter requested but cache/v/sizes missing", name);
    } else {
        NV_REQUIRE(k_out, NV_ERR_INVALID_ARG, "%s: k_out is NULL and no scatter requested", name);
    }
    NV_REQUIRE(q_stride_t % 8 == 0 && k_stride_t % 8 == 0 && (!store || v_stride_t % 8 == 0),
               NV_ERR_UNSUPPORTED, "%s: token strides must be multiples of 8 elements", name);
    NV_REQUIRE(q_stride_t >= (int64_t)H_q * D && k_stride_t >= (int64_t)H_kv * D, NV_ERR_INVALID_ARG,
               "%s: token stride smaller than a row", name);
    NV_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)q_w | (uintptr_t)k_w |
                (uintptr_t)cos_sin | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)q_out |
                (uintptr_t)k_out) % 16 == 0,
               NV_ERR_INVALID_ARG, "%s: pointers must be 16-byte aligned", name);
    const int64_t num_slots = store ? num_blocks * (int64_t)block_size : 0;
    hipStream_t st = (hipStream_t)stream;
#define NV_QKR_ARGS                                                                              \
    (const uint4*)q, (const uint4*)k, (const uint4*)(store ? v : nullptr), (const uint4*)q_w,    \
        (const uint4*)k_w, (const float4*)cos_sin, positions, slot_mapping, (uint4*)k_cache,      \
        (uint4*)v_cache, (uint4*)q_out, (uint4*)k_out, H_q, H_kv, max_pos, eps, num_slots,        \
        q_stride_t / 8, k_stride_t / 8, v_stride_t / 8
    if (q_w) hipLaunchKernelGGL(qknorm_rope_store_kernel<true>, dim3(T), dim3(256), 0, st, NV_QKR_ARGS);
    else hipLaunchKernelGGL(qknorm_rope_store_kernel<false>, dim3(T), dim3(256), 0, st, NV_QKR_ARGS);
#undef NV_QKR_ARGS
    return check_launch(name);
}

extern "C" int nv_rope(const void* q, const void* k, void* q_out, void* k_out, const float* cos_sin,
                       const int64_t* positions, int T, int H_q, int H_kv, int D, int64_t max_pos,
                       int64_t q_stride_t, int64_t k_stride_t, void* stream) {
    return qknorm_rope_store_impl("nv_rope", q, k, nullptr, nullptr, nullptr, cos_sin, positions,
                                  nullptr, nullptr, nullptr, q_out, k_out, T, H_q, H_kv, D, max_pos,
                                  0.f, 0, 0, q_stride_t, k_stride_t, 0, stream);
}

extern "C" int nv_qknorm_rope_store(const void* q, const void* k, const void* v, const void* q_w,
                                    const void* k_w, const float* cos_sin, const int64_t* positions,
                                    const int32_t* slot_mapping, void* k_cache, void* v_cache,
                                    void* q_out, void* k_out, int T, int H_q, int H_kv, int D,
                                    int64_t max_pos, float eps, int64_t num_blocks, int block_size,
                                    int64_t q_stride_t, int64_t k_stride_t, int64_t v_stride_t,
                                    void* stream) {
    return qknorm_rope_store_impl("nv_qknorm_rope_store", q, k, v, q_w, k_w, cos_sin, positions,
                                  slot_mapping, k_cache, v_cache, q_out, k_out, T, H_q, H_kv, D,
                                  max_pos, eps, num_blocks, block_size, q_stride_t, k_stride_t,
                                  v_stride_t, stream);
}
