// Varlen prefill / chunked-prefill attention over the paged KV cache for gfx950.
//
// Flash-style: one workgroup (4 waves) owns a 128-row query tile of one (sequence, query
// head); each wave owns 32 query rows.  K/V tiles of 64 keys are gathered through the
// block table into a double-buffered, swizzled LDS image by LDS-DMA and consumed by
// bf16 MFMA 32x32x16:
//   S^T = K . Q^T      (A = K rows from LDS via ds_read_b128, B = Q fragments in registers)
//   O^T += V^T . P^T   (A = V^T via ds_read_b64_tr_b16 transposed reads, B = P^T taken
//                       straight from the S^T accumulator registers, converted to bf16)
// With the "swapped" product the 32 scores of a query row live in the registers of two
// lanes (l, l^32), so the online softmax needs one cross-lane exchange per tile.
//
// Mask: causal, aligned to the bottom-right corner (query i of n_q sees keys
// j <= n_k - n_q + i), which is what chunked prefill needs (n_k = computed + n_q).
// Semantics: flash_attn_varlen_func(..., causal=True, block_table=...) as called at
// nanovllm/layers/attention.py:241-252 with metadata from nanovllm/engine/model_runner.py:85-119.
#include "nv_common.h"

namespace nv {

constexpr int kBM = 128;  // query rows per workgroup
constexpr int kBN = 64;   // keys per tile
constexpr float kMasked = -1.0e30f;

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct PrefillParams {
    const uint4* q;
    const uint4* k_cache;
    const uint4* v_cache;
    __bf16* o;
    const int32_t* cu_q;
    const int32_t* cu_k;
    const int32_t* block_tables;
    int B, max_blocks, H_q, H_kv, block_size, n_qtiles;
    int64_t num_blocks;
    int64_t q_stride_vec, o_stride;
    float scale_log2;
    int skip_single;
};

// LDS image of a [64 keys][128 dims] bf16 tile: 8-row x 32-column subtiles of 512 bytes,
// with the 16-byte chunk inside a subtile row XOR-ed by bits 2..3 of the row.  Byte offset
// of 16-byte chunk `ch` (0..15) of row `row`.  Conflict-free for the ds_read_b128 row
// reads of the K tile and the transposed reads of the V tile, and every read address of
// a lane is one of two lane constants plus a compile-time immediate.
__device__ __forceinline__ int tile_off(int row, int ch) {
    return 2048 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(const PrefillParams p) {
    // [2 buffers][K tile 16 KiB | V tile 16 KiB]; the only LDS object of the kernel
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 32768];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    const int b = blockIdx.z;
    const int hq = blockIdx.y;
    const int qt = p.n_qtiles - 1 - blockIdx.x;  // heaviest (latest) tiles first
    const int q_begin = p.cu_q[b];
    const int n_q = p.cu_q[b + 1] - q_begin;
    const int n_k = p.cu_k[b + 1] - p.cu_k[b];
    if (n_q <= 0 || qt * kBM >= n_q) return;
    if (p.skip_single && n_q == 1) return;
    const int hkv = hq / (p.H_q / p.H_kv);
    const int shift = n_k - n_q;            // query i sees keys <= i + shift
    const int q0 = qt * kBM;                // first query row of the workgroup
    const int rw0 = q0 + wave * 32;         // first query row of the wave
    const bool wave_live = rw0 < n_q;
    const int last_row = min(q0 + kBM, n_q) - 1;
    const int max_key = min(last_row + shift, n_k - 1);  // last key any row here can see
    const int nt = max_key >= 0 ? max_key / kBN + 1 : 0;

    // ---- Q fragments: B operand of S^T, lane (r, h) holds Q[row r][16s + 8h .. +7] ----
    bf16x8 qf[8];
    {
        const int row = min(rw0 + r, n_q - 1);
        const uint4* qp = p.q + (int64_t)(q_begin + row) * p.q_stride_vec + (int64_t)hq * 16;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = __builtin_bit_cast(bf16x8, qp[2 * s + h]);
    }

    f32x16 oacc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    float m_run = kMasked;  // running max (log2 domain), shared by lanes l and l^32
    float l_run = 0.f;      // this lane's share of the running denominator

    // ---- staging: K/V tiles go global -> LDS by LDS-DMA (no registers), double-buffered.
    // One DMA wave-instruction fills 1 KiB of LDS linearly (lane l -> byte 16 l), i.e. two
    // adjacent 512-byte subtiles; each lane picks the global (row, chunk) that the swizzled
    // image expects at its LDS position.  Wave w issues pieces 4w .. 4w+3 of K and of V.
    const int32_t* bt = p.block_tables + (int64_t)b * p.max_blocks;
    // Block-table entries are fetched one tile ahead of the DMA that needs them (a vector
    // load: the compiler will not use a scalar load next to LDS-DMA writes), so the DMA
    // issue right after the barrier never waits on the table.
    const int tiles_per_block = p.block_size / kBN;
    auto table_entry = [&](int t) { return bt[t / tiles_per_block]; };
    auto stage = [&](int t, int buf, int phys_entry) {
        const int key0 = t * kBN;
        // block_size is a multiple of 64 here, so a tile sits inside one cache block
        const int off = (t % tiles_per_block) * kBN;
        int64_t phys = __builtin_amdgcn_readfirstlane(phys_entry);
        phys = phys < 0 ? 0 : (phys >= p.num_blocks ? p.num_blocks - 1 : phys);
        const int64_t base = ((phys * p.block_size + off) * p.H_kv + hkv) * 16;
        const int last = n_k - 1 - key0;  // last valid row of this tile (>= 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = wave * 4 + i;
            int row = (pc >> 1) * 8 + ((lane & 31) >> 2);
            const int ch = (2 * (pc & 1) + (lane >> 5)) * 4 + ((lane & 3) ^ ((row >> 2) & 3));
            // rows beyond the sequence re-read the last valid row: finite data, masked scores
            row = row <= last ? row : last;
            const int64_t vec = base + (int64_t)row * (p.H_kv * 16) + ch;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(p.k_cache + vec),
                (__attribute__((address_space(3))) void*)(lds + buf * 32768 + pc * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(p.v_cache + vec),
                (__attribute__((address_space(3))) void*)(lds + buf * 32768 + 16384 + pc * 1024), 16, 0, 0);
        }
    };

    int phys_next = 0;  // table entry of tile t+1 while tile t is computed
    if (nt > 0) stage(0, 0, table_entry(0));
    if (nt > 1) phys_next = table_entry(1);

    for (int t = 0; t < nt; ++t) {
        // tile t has landed for every wave, and every wave is done reading the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int phys_use = phys_next;
        if (t + 2 < nt) phys_next = table_entry(t + 2);
        if (t + 1 < nt) stage(t + 1, (t + 1) & 1, phys_use);
        const unsigned char* ldsK = lds + (t & 1) * 32768;
        const unsigned char* ldsV = ldsK + 16384;
        const int kt = t * kBN;
        // wave-uniform: does this wave have any row that sees a key of this tile?
        const bool active = wave_live && kt <= min(rw0 + 31, n_q - 1) + shift;
        if (active) {
            // ---------------- S^T = K . Q^T : 2 key blocks of 32 ----------------
            f32x16 sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
                const int row = kb * 32 + r;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const bf16x8 a = __builtin_bit_cast(
                        bf16x8, *reinterpret_cast<const uint4*>(ldsK + tile_off(row, 2 * s + h)));
                    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], sacc[kb], 0, 0, 0);
                }
            }
            // ---------------- online softmax over the 64 keys ----------------
            // register i of block kb is key kt + kb*32 + (i&3) + 8*(i>>2) + 4h; query row rw0 + r
            const int qpos = rw0 + r + shift;  // last visible key of this lane's query row
            const bool need_mask = kt + kBN - 1 > rw0 + shift;  // wave-uniform
            float tmax = kMasked;
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = kt + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        const float s = key <= qpos ? sacc[kb][i] * p.scale_log2 : kMasked;
                        sacc[kb][i] = s;
                        tmax = fmaxf(tmax, s);
                    }
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float s = sacc[kb][i] * p.scale_log2;
                        sacc[kb][i] = s;
                        tmax = fmaxf(tmax, s);
                    }
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            float psum = 0.f;
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float s = sacc[kb][i];
                        const float e = s > 0.5f * kMasked ? __builtin_amdgcn_exp2f(s - m_new) : 0.f;
                        sacc[kb][i] = e;
                        psum += e;
                    }
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float e = __builtin_amdgcn_exp2f(sacc[kb][i] - m_new);
                        sacc[kb][i] = e;
                        psum += e;
                    }
            }
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;

            // ---------------- O^T += V^T . P^T : 4 k-steps of 16 keys ----------------
            const int i16 = lane & 15;
            const int g16 = (lane >> 4) & 1;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kb = s >> 1;
                const int s2 = s & 1;
                bf16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (__bf16)sacc[kb][8 * s2 + j];
                // keys of element j: kb*32 + 16*s2 + 8*(j>>2) + 4h + (j&3)
                const int row1 = kb * 32 + 16 * s2 + 4 * h + (i16 >> 2);
                const int row2 = row1 + 8;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int ch = d * 4 + 2 * g16 + ((i16 & 3) >> 1);
                    const int sub = 8 * (i16 & 1);
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_s16x4*)(ldsV + tile_off(row1, ch) + sub));
                    const s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_s16x4*)(ldsV + tile_off(row2, ch) + sub));
                    s16x8 va;
                    va[0] = v1[0]; va[1] = v1[1]; va[2] = v1[2]; va[3] = v1[3];
                    va[4] = v2[0]; va[5] = v2[1]; va[6] = v2[2]; va[7] = v2[3];
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va),
                                                                      pb, oacc[d], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: O = O^T / l ; lane (r, h) holds dims d*32 + 8*(i>>2) + 4h + (i&3) ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const int row = rw0 + r;
    if (!wave_live || row >= n_q) return;
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    __bf16* op = p.o + (int64_t)(q_begin + row) * p.o_stride + (int64_t)hq * kHeadDim;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack_bf16(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv);
            w.y = pack_bf16(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(op + d * 32 + 8 * g + 4 * h) = w;
        }
}

}  // namespace nv

extern "C" int nv_attn_prefill_varlen(const void* q, const void* k_cache, const void* v_cache,
                                      void* o, const int32_t* cu_seqlens_q,
                                      const int32_t* cu_seqlens_k, const int32_t* block_tables,
                                      int B, int max_blocks, int max_seqlen_q, int max_seqlen_k,
                                      int H_q, int H_kv, int D, int64_t num_blocks, int block_size,
                                      float scale, int64_t q_stride_t, int64_t o_stride_t,
                                      int skip_single_query, void* stream) {
    using namespace nv;
    NV_REQUIRE(B >= 0 && max_blocks > 0 && max_seqlen_q >= 0 && max_seqlen_k >= 0 && H_q > 0 &&
                   H_kv > 0 && num_blocks > 0 && block_size > 0,
               NV_ERR_INVALID_ARG, "nv_attn_prefill_varlen: bad sizes B=%d max_q=%d max_k=%d", B,
               max_seqlen_q, max_seqlen_k);
    NV_REQUIRE(D == kHeadDim, NV_ERR_UNSUPPORTED, "nv_attn_prefill_varlen: head_dim %d (128 only)", D);
    NV_REQUIRE(H_q % H_kv == 0, NV_ERR_INVALID_ARG, "nv_attn_prefill_varlen: H_q %% H_kv != 0");
    NV_REQUIRE(block_size % kBN == 0, NV_ERR_UNSUPPORTED,
               "nv_attn_prefill_varlen: block_size %d must be a multiple of %d", block_size, kBN);
    if (B == 0 || max_seqlen_q == 0) return NV_OK;
    NV_REQUIRE(q && k_cache && v_cache && o && cu_seqlens_q && cu_seqlens_k && block_tables,
               NV_ERR_INVALID_ARG, "nv_attn_prefill_varlen: null pointer");
    NV_REQUIRE(q_stride_t % 8 == 0 && o_stride_t % 4 == 0 && q_stride_t >= (int64_t)H_q * D &&
                   o_stride_t >= (int64_t)H_q * D,
               NV_ERR_INVALID_ARG, "nv_attn_prefill_varlen: bad token strides");
    NV_REQUIRE(((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)o) % 16 == 0,
               NV_ERR_INVALID_ARG, "nv_attn_prefill_varlen: pointers must be 16-byte aligned");
    NV_REQUIRE((int64_t)max_blocks * block_size >= max_seqlen_k, NV_ERR_INVALID_ARG,
               "nv_attn_prefill_varlen: block table too narrow for max_seqlen_k");
    NV_REQUIRE(H_q <= 65535 && B <= 65535, NV_ERR_UNSUPPORTED,
               "nv_attn_prefill_varlen: H_q and B must be <= 65535");

    PrefillParams p;
    p.q = (const uint4*)q;
    p.k_cache = (const uint4*)k_cache;
    p.v_cache = (const uint4*)v_cache;
    p.o = (__bf16*)o;
    p.cu_q = cu_seqlens_q;
    p.cu_k = cu_seqlens_k;
    p.block_tables = block_tables;
    p.B = B;
    p.max_blocks = max_blocks;
    p.H_q = H_q;
    p.H_kv = H_kv;
    p.block_size = block_size;
    p.n_qtiles = (max_seqlen_q + kBM - 1) / kBM;
    p.num_blocks = num_blocks;
    p.q_stride_vec = q_stride_t / 8;
    p.o_stride = o_stride_t;
    p.scale_log2 = scale * 1.4426950408889634f;
    p.skip_single = skip_single_query;
    hipLaunchKernelGGL(attn_prefill_kernel, dim3(p.n_qtiles, H_q, B), dim3(256), 0,
                       (hipStream_t)stream, p);
    return check_launch("nv_attn_prefill_varlen");
}
