// KV-cache scatter: one workgroup per token copies the token's K row and V row
// (H_kv*D bf16 each, 2 KiB for H_kv=8, D=128) into the paged cache with 16-byte
// vector accesses.  HBM-bound byte copy, bit-exact.
// Semantics: Attention.store_kvcache, nanovllm/layers/attention.py:210-219.
#include "nv_common.h"

namespace nv {

__global__ __launch_bounds__(256) void store_kvcache_kernel(
    const uint4* __restrict__ k, const uint4* __restrict__ v, uint4* __restrict__ k_cache,
    uint4* __restrict__ v_cache, const int32_t* __restrict__ slot_mapping, int row_vec,
    int64_t k_stride_vec, int64_t v_stride_vec, int64_t num_slots) {
    const int t = blockIdx.x;
    const int64_t slot = slot_mapping[t];
    if (slot < 0 || slot >= num_slots) return;  // padding slot: nothing to write
    const uint4* ksrc = k + (int64_t)t * k_stride_vec;
    const uint4* vsrc = v + (int64_t)t * v_stride_vec;
    uint4* kdst = k_cache + slot * row_vec;
    uint4* vdst = v_cache + slot * row_vec;
    for (int i = threadIdx.x; i < 2 * row_vec; i += blockDim.x) {
        if (i < row_vec) kdst[i] = ksrc[i];
        else vdst[i - row_vec] = vsrc[i - row_vec];
    }
}

}  // namespace nv

extern "C" int nv_store_kvcache(const void* k, const void* v, void* k_cache, void* v_cache,
                                const int32_t* slot_mapping, int T, int H_kv, int D,
                                int64_t num_blocks, int block_size, int64_t k_stride_t,
                                int64_t v_stride_t, void* stream) {
    using namespace nv;
    NV_REQUIRE(T >= 0 && H_kv > 0 && D > 0 && num_blocks > 0 && block_size > 0, NV_ERR_INVALID_ARG,
               "nv_store_kvcache: bad sizes T=%d H_kv=%d D=%d num_blocks=%lld block_size=%d", T,
               H_kv, D, (long long)num_blocks, block_size);
    if (T == 0) return NV_OK;
    NV_REQUIRE(k && v && k_cache && v_cache && slot_mapping, NV_ERR_INVALID_ARG,
               "nv_store_kvcache: null pointer");
    const int64_t row = (int64_t)H_kv * D;  // elements per token row
    NV_REQUIRE(row % 8 == 0 && k_stride_t % 8 == 0 && v_stride_t % 8 == 0, NV_ERR_UNSUPPORTED,
               "nv_store_kvcache: row (%lld) and token strides must be multiples of 8 elements",
               (long long)row);
    NV_REQUIRE(k_stride_t >= row && v_stride_t >= row, NV_ERR_INVALID_ARG,
               "nv_store_kvcache: token stride smaller than a row");
    NV_REQUIRE(((uintptr_t)k | (uintptr_t)v | (uintptr_t)k_cache | (uintptr_t)v_cache) % 16 == 0,
               NV_ERR_INVALID_ARG, "nv_store_kvcache: pointers must be 16-byte aligned");
    const int row_vec = (int)(row / 8);
    hipLaunchKernelGGL(store_kvcache_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)k, (const uint4*)v, (uint4*)k_cache, (uint4*)v_cache,
                       slot_mapping, row_vec, k_stride_t / 8, v_stride_t / 8,
                       num_blocks * (int64_t)block_size);
    return check_launch("nv_store_kvcache");
}
