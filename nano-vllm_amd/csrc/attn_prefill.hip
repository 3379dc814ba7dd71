// Varlen prefill / chunked-prefill attention over the paged KV cache for gfx950.
//
// Flash-style bf16 MFMA kernel.  A workgroup is 4 waves; each wave owns 32 query rows of one
// query head.  The waves of a workgroup share one kv head: they are the `hpw` = gcd(G, 4)
// query heads of a GQA group times 4/hpw consecutive 32-row blocks, so a K/V tile staged in
// LDS once serves up to 4 query heads.  K/V tiles of 64 keys are gathered through the block
// table into a double-buffered, swizzled LDS image by LDS-DMA and consumed by
// v_mfma_f32_32x32x16_bf16:
//   S^T = K . Q^T      (A = K rows from LDS via ds_read_b128, B = Q fragments in registers)
//   O^T += V^T . P^T   (A = V^T via ds_read_b64_tr_b16 transposed reads, B = P^T taken
//                       straight from the S^T accumulator registers, converted to bf16)
// With the "swapped" product the 64 scores of a query row live in the registers of two
// lanes (l, l^32), so the online softmax needs one cross-lane exchange per tile.  The
// running maximum is only raised when a tile exceeds it by more than kRescaleThreshold
// (log2 units), which removes the accumulator rescale from almost every tile.
//
// Workgroups that read the same (sequence, kv head) run on the same XCD (ids congruent
// mod 8 share an XCD under round-robin dispatch), so K/V re-reads hit that XCD's L2.
//
// Mask: causal, aligned to the bottom-right corner (query i of n_q sees keys
// j <= n_k - n_q + i), which is what chunked prefill needs (n_k = computed + n_q).
// Semantics: flash_attn_varlen_func(..., causal=True, block_table=...) as called at
// nanovllm/layers/attention.py:241-252 with metadata from nanovllm/engine/model_runner.py:85-119.
#include "nv_common.h"
#include "attn_prefill_shared.h"

#ifndef NV_PF_MAX_ITEMS
#define NV_PF_MAX_ITEMS 4
#endif

namespace nv {

// LDS image of a [64 keys][128 dims] bf16 tile: 8-row x 32-column subtiles of 512 bytes,
// with the 16-byte chunk inside a subtile row XOR-ed by bits 2..3 of the row.  Byte offset
// of 16-byte chunk `ch` (0..15) of row `row`.  Conflict-free for the ds_read_b128 row
// reads of the K tile and the transposed reads of the V tile, and every read address of
// a lane is one of two lane constants plus a compile-time immediate.
// Lanes l and l ^ 32 exchange a value with one VALU instruction (v_permlane32_swap) instead of
// a round trip through the LDS crossbar (ds_bpermute), which sits on the critical path of
// every tile's softmax.
__device__ __forceinline__ float max_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

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

    // ---- workgroup -> (sequence, kv head, head group, query tile), XCD-aware ----
    const int split = blockIdx.x / p.grid_base;          // key range of this workgroup
    const int bid = blockIdx.x - split * p.grid_base;
    const int xcd = bid & 7;
    const int slot = bid >> 3;
    const int round = slot / p.slots_per_pair;          // 8 >> xs_shift pairs per round of XCDs
    const int pair = round * (8 >> p.xs_shift) + (xcd >> p.xs_shift);
    const int widx = ((slot - round * p.slots_per_pair) << p.xs_shift) + (xcd & ((1 << p.xs_shift) - 1));
    if (pair >= p.n_pairs || widx >= p.wg_per_pair) return;
    const int groups = p.wg_per_pair / p.n_qtiles;      // head groups per kv head (G / hpw)
    const int hg = widx % groups;
    const int qt = p.n_qtiles - 1 - widx / groups;      // heaviest (latest) tiles first
    const int b = pair / p.H_kv;
    const int hkv = pair - b * p.H_kv;
    const int G = p.H_q / p.H_kv;
    const int hq = hkv * G + hg * p.hpw + wave % p.hpw;

    const int q_begin = p.cu_q[b];
    const int n_q = p.cu_q[b + 1] - q_begin;
    const int n_k = p.cu_k[b + 1] - p.cu_k[b];
    const int q0 = qt * p.rows_per_wg;                  // first query row of the workgroup
    if (n_q <= 0 || q0 >= n_q) return;
    if (p.skip_single && n_q == 1) return;
    const int shift = n_k - n_q;                        // query i sees keys <= i + shift
    const int rw0 = q0 + (wave / p.hpw) * 32;           // first query row of the wave
    const bool wave_live = rw0 < n_q;
    const int last_row = min(q0 + p.rows_per_wg, n_q) - 1;
    const int max_key = min(last_row + shift, n_k - 1);  // last key any row here can see
    const int nt_all = max_key >= 0 ? max_key / kBN + 1 : 0;
    const int t_begin = split * p.tiles_per_split;
    const int nt = min(nt_all, t_begin + p.tiles_per_split);  // this split covers tiles t_begin .. nt-1

    // ---- Q fragments: B operand of S^T, lane (r, h) holds Q[row r][16s + 8h .. +7] ----
    bf16x8 qf[8];
    {
        const int row = min(rw0 + r, n_q - 1);
        const uint4* qp = p.q + (int64_t)(q_begin + row) * p.q_stride_vec + (int64_t)hq * 16;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = __builtin_bit_cast(bf16x8, qp[2 * s + h]);
    }
    // The loop masks raw scores with -inf and takes their maximum before scaling, which needs a positive
    // scale: a negative one moves its sign into Q (exact), a zero one clears Q and scales by 1 (every visible
    // score 0, every hidden one still -inf).
    float scale_log2 = p.scale_log2;
    if (scale_log2 <= 0.f) {  // uniform, never taken by a softmax scale of 1/sqrt(d)
        const uint32_t flip = scale_log2 < 0.f ? 0x80008000u : 0u, keep = scale_log2 < 0.f ? ~0u : 0u;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            uint4 w = __builtin_bit_cast(uint4, qf[s]);
            w.x = (w.x ^ flip) & keep; w.y = (w.y ^ flip) & keep; w.z = (w.z ^ flip) & keep; w.w = (w.w ^ flip) & keep;
            qf[s] = __builtin_bit_cast(bf16x8, w);
        }
        scale_log2 = scale_log2 < 0.f ? -scale_log2 : 1.f;
    }

    f32x16 oacc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    float m_run = kRunInit;  // reference max of the exponentials (log2 domain), lanes l and l^32 agree
    float l_run = 0.f;       // this lane's share of the running denominator

    // ---- staging: K/V tiles go global -> LDS by LDS-DMA (no registers), double-buffered.
    // One DMA wave-instruction fills 1 KiB of LDS linearly (lane l -> byte 16 l), i.e. two
    // adjacent 512-byte subtiles; each lane picks the global (row, chunk) that the swizzled
    // image expects at its LDS position.  Wave w issues pieces 4w .. 4w+3 of K and of V.
    const int32_t* bt = p.block_tables + (int64_t)b * p.max_blocks;
    // Block-table entries are fetched one tile ahead of the DMA that needs them (a vector
    // load: the compiler will not use a scalar load next to LDS-DMA writes), so the DMA
    // issue right after the barrier never waits on the table.
    const int tiles_per_block = p.block_size / kBN;
    const int row_vec = p.H_kv * 16;                                   // uint4 per cache token row
    const int64_t block_vec = (int64_t)p.block_size * row_vec;         // uint4 per cache block
    const int max_block = (int)(p.num_blocks - 1 < 0x7fffffff ? p.num_blocks - 1 : 0x7fffffff);
    // Per-lane part of a DMA source address, constant over the tiles: piece i of this wave
    // covers cache rows (pc>>1)*8 .. +7 and four 16-byte chunks per row, picked so that the
    // linear LDS fill produces the swizzled image.  In bytes relative to the tile base (a tile
    // spans 64 cache rows, far below 4 GiB).
    int dma_row[4];
    uint32_t dma_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pc = wave * 4 + i;
        dma_row[i] = (pc >> 1) * 8 + ((lane & 31) >> 2);
        const int ch = (2 * (pc & 1) + (lane >> 5)) * 4 + ((lane & 3) ^ ((dma_row[i] >> 2) & 3));
        dma_off[i] = (uint32_t)(dma_row[i] * row_vec + ch) * 16u;   // bytes
    }
    // tile t -> buffer BUF (compile-time, so every LDS address below is a lane constant plus
    // an immediate)
    // block_size is a multiple of 64 here, so a tile sits inside one cache block: it is named by
    // its table entry and its key offset `off` inside the block; `last` = last valid row of the tile.
    auto stage = [&](auto buf_tag, int phys_entry, int off, int last) {
        constexpr int BUF = decltype(buf_tag)::value;
        int phys = __builtin_amdgcn_readfirstlane(phys_entry);
        phys = phys < 0 ? 0 : (phys > max_block ? max_block : phys);
        const int64_t base = (int64_t)phys * block_vec + off * row_vec + hkv * 16;
        const char* kbase = reinterpret_cast<const char*>(p.k_cache + base);   // wave-uniform
        const char* vbase = reinterpret_cast<const char*>(p.v_cache + base);
        uint32_t o[4];
        if (last >= kBN - 1) {  // wave-uniform: every row of the tile exists
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = dma_off[i];
        } else {
            // rows beyond the sequence re-read the last valid row: finite data, masked scores
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = dma_off[i] - (uint32_t)(max(dma_row[i] - last, 0) * row_vec) * 16u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = wave * 4 + i;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(kbase + o[i]),
                (__attribute__((address_space(3))) void*)(lds + BUF * 32768 + pc * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(vbase + o[i]),
                (__attribute__((address_space(3))) void*)(lds + BUF * 32768 + 16384 + pc * 1024), 16, 0, 0);
        }
    };
    using buf0 = std::integral_constant<int, 0>;
    using buf1 = std::integral_constant<int, 1>;

    // tile t_begin + j lives in buffer j & 1
    // Tile bookkeeping is incremental (no divisions in the loop): st_* names the tile staged next
    // (t + 1), pf_* the tile whose table entry is fetched next (t + 2).
    int phys_next = 0;  // table entry of tile t+1 while tile t is computed
    if (t_begin < nt)
        stage(buf0{}, bt[t_begin / tiles_per_block], (t_begin % tiles_per_block) * kBN, n_k - 1 - t_begin * kBN);
    if (t_begin + 1 < nt) phys_next = bt[(t_begin + 1) / tiles_per_block];
    int st_off = ((t_begin + 1) % tiles_per_block) * kBN;
    int st_last = n_k - 1 - (t_begin + 1) * kBN;
    int pf_blk = (t_begin + 2) / tiles_per_block;
    int pf_off = ((t_begin + 2) % tiles_per_block) * kBN;

    const int i16 = lane & 15;
    const int g16 = (lane >> 4) & 1;
    const int qpos = rw0 + r + shift;               // last visible key of this lane's query row
    const int wave_last_key = min(rw0 + 31, n_q - 1) + shift;  // last key any row of the wave sees

    // One tile: buffer BUF holds tile t, the other buffer receives tile t + 1.
    auto tile_step = [&](auto buf_tag, int t) {
        constexpr int BUF = decltype(buf_tag)::value;
        // tile t has landed for every wave, and every wave is done reading the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int phys_use = phys_next;
        if (t + 2 < nt) phys_next = bt[pf_blk];
        if (t + 1 < nt) stage(std::integral_constant<int, 1 - BUF>{}, phys_use, st_off, st_last);
        st_off = st_off + kBN == p.block_size ? 0 : st_off + kBN;
        st_last -= kBN;
        pf_off += kBN;
        if (pf_off == p.block_size) {
            pf_off = 0;
            ++pf_blk;
        }
        const unsigned char* ldsK = lds + BUF * 32768;
        const unsigned char* ldsV = ldsK + 16384;
        const int kt = t * kBN;
        // wave-uniform: does this wave have any row that sees a key of this tile?
        if (!(wave_live && kt <= wave_last_key)) return;

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
        // ---------------- online softmax over the 64 keys (raw scores, scale folded in) ----
        // register i of block kb is key kt + kb*32 + (i&3) + 8*(i>>2) + 4h; query row rw0 + r
        if (kt + kBN - 1 > rw0 + shift) {  // wave-uniform: the tile touches the diagonal
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    sacc[kb][i] = key <= qpos ? sacc[kb][i] : -INFINITY;
                }
        }
        float tmax = sacc[0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, sacc[0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, sacc[1][i]);
        tmax = max_xor32(tmax) * scale_log2;
        if (!__all(tmax <= m_run + kRescaleThreshold)) {  // rare after the first tiles
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kb][i], scale_log2, -m_run));
                sacc[kb][i] = e;
                psum += e;
            }
        l_run += psum;

        // ---------------- O^T += V^T . P^T : 4 k-steps of 16 keys ----------------
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
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), pb,
                                                                  oacc[d], 0, 0, 0);
            }
        }
    };

    for (int t = t_begin; t < nt; t += 2) {
        tile_step(buf0{}, t);
        if (t + 1 < nt) tile_step(buf1{}, t + 1);
    }

    // ---- epilogue ----
    // Lane (r, h) holds row rw0 + r, dims d*32 + 8*(i>>2) + 4h + (i&3) of O^T: stored as it
    // stands every store instruction would touch 32 rows.  The tile buffers are free now,
    // so each wave transposes through its own LDS area (32 rows x 256 B, row stride 272 B:
    // conflict-free both ways) and writes whole 256-byte row segments, 4 rows per instruction.
    const float l_tot = sum_xor32(l_run);
    __syncthreads();  // every wave is done reading K/V tiles
    if (!wave_live) return;
    unsigned char* stg = lds + wave * (32 * 272);
    const int rows_here = n_q - rw0;  // rows of this wave that exist (>= 1)
    auto flush = [&](unsigned char* gbase, int64_t row_bytes) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rr = 4 * it + (lane >> 4);
            const uint4 v = *reinterpret_cast<const uint4*>(stg + rr * 272 + (lane & 15) * 16);
            if (rr < rows_here) *reinterpret_cast<uint4*>(gbase + rr * row_bytes + (lane & 15) * 16) = v;
        }
    };
    if (p.nsplit > 1) {
        // unnormalised state of this key range; an empty range leaves l = 0 and is skipped
        // by the merge, so its accumulators need not be written
        float* dst0 = p.part + (((int64_t)(q_begin + rw0) * p.H_q + hq) * p.nsplit + split) * kPartStride;
        const int64_t row_bytes = (int64_t)p.H_q * p.nsplit * kPartStride * 4;
        if (h == 0 && r < rows_here) {
            float* dst = dst0 + (int64_t)r * (row_bytes / 4);
            dst[kHeadDim] = m_run;
            dst[kHeadDim + 1] = l_tot;
        }
        if (t_begin >= nt) return;
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // 64 fp32 dims = 256 B per pass
#pragma unroll
            for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = 2 * half + dd;
                    *reinterpret_cast<float4*>(stg + r * 272 + (dd * 32 + 8 * g + 4 * h) * 4) =
                        make_float4(oacc[d][4 * g], oacc[d][4 * g + 1], oacc[d][4 * g + 2], oacc[d][4 * g + 3]);
                }
            flush(reinterpret_cast<unsigned char*>(dst0) + half * 256, row_bytes);
        }
        return;
    }
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack_bf16(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv);
            w.y = pack_bf16(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(stg + r * 272 + (d * 32 + 8 * g + 4 * h) * 2) = w;
        }
    flush(reinterpret_cast<unsigned char*>(p.o + (int64_t)(q_begin + rw0) * p.o_stride + (int64_t)hq * kHeadDim),
          p.o_stride * 2);
}

// Combine the key-range states of one query row: block = 8 (row, head) pairs x 32 threads
// (4 dims each).  grid = (ceil(max_seqlen_q * H_q / 8), B).
__global__ __launch_bounds__(256) void attn_prefill_merge_kernel(const PrefillParams p) {
    const int b = blockIdx.y;
    const int pair = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int row = pair / p.H_q;
    const int d4 = threadIdx.x & 31;
    const int q_begin = p.cu_q[b];
    const int n_q = p.cu_q[b + 1] - q_begin;
    if (row >= n_q || (p.skip_single && n_q == 1)) return;
    {
        const int hq = pair - row * p.H_q;
        const float* base = p.part + ((int64_t)(q_begin + row) * p.H_q + hq) * p.nsplit * kPartStride;
        float mmax = kRunInit;
        for (int s = 0; s < p.nsplit; ++s) {
            const float2 ml = *reinterpret_cast<const float2*>(base + s * kPartStride + kHeadDim);
            if (ml.y > 0.f) mmax = fmaxf(mmax, ml.x);
        }
        float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
        float den = 0.f;
        for (int s = 0; s < p.nsplit; ++s) {
            const float2 ml = *reinterpret_cast<const float2*>(base + s * kPartStride + kHeadDim);
            if (!(ml.y > 0.f)) continue;
            const float4 a = *reinterpret_cast<const float4*>(base + s * kPartStride + 4 * d4);
            const float f = __builtin_amdgcn_exp2f(ml.x - mmax);
            num.x += a.x * f;
            num.y += a.y * f;
            num.z += a.z * f;
            num.w += a.w * f;
            den += ml.y * f;
        }
        const float r = den > 0.f ? 1.0f / den : 0.f;
        uint2 w;
        w.x = pack_bf16(num.x * r, num.y * r);
        w.y = pack_bf16(num.z * r, num.w * r);
        __bf16* o = p.o + (int64_t)(q_begin + row) * p.o_stride + (int64_t)hq * kHeadDim;
        *reinterpret_cast<uint2*>(o + 4 * d4) = w;
    }
}

constexpr int kMaxItemsPerWg = NV_PF_MAX_ITEMS;   // query tiles per workgroup of the hand-scheduled kernel (<= 8)

static int gcd_int(int a, int b) { return b == 0 ? a : gcd_int(b, a % b); }

struct PrefillShape;
static void prefill_set_grid(PrefillShape& sh);

struct PrefillShape {
    int nw, hpw, rows_per_wg, n_qtiles, wg_per_pair, n_pairs;
    int xs_shift, slots_per_pair;   // see PrefillParams
    int64_t grid_base;
    int nsplit, tiles_per_split;
    bool hand_scheduled;   // attn_prefill_asm_kernel (64 query rows per wave, never split) instead of attn_prefill_kernel
    int items_per_wg;      // its workgroups take 1, 2 or 4 query tiles each (chosen so that all have equal work)
};

// Workgroups per key range: whole rounds of 8 XCDs; with fewer than 8 pairs each pair takes 8, 4 or 2 of them.
static void prefill_set_grid(PrefillShape& sh) {
    // the fullest XCD would hold over 15 % more pairs than the average (6 or 12 pairs on 8 XCDs, one on 8): spread
    // each pair over the fewest XCDs that make the count even again (its K/V tiles are then read into that many L2s,
    // which a compute-bound launch does not notice).  One 8192-token sequence on a one-kv-head shard 695 -> 170 us,
    // three 4096-token sequences x 4 kv heads 457 -> 343; 7 pairs stay as they are (one idle XCD of eight costs
    // less than the spread: 237 against 247 us)
    sh.xs_shift = 0;
    if (((sh.n_pairs + 7) / 8) * 8 * 20 > sh.n_pairs * 23)
        while (sh.xs_shift < 3 && ((sh.n_pairs << sh.xs_shift) & 7) != 0) ++sh.xs_shift;
    const int xs = 1 << sh.xs_shift, per_round = 8 >> sh.xs_shift;
    sh.slots_per_pair = (sh.wg_per_pair + xs - 1) / xs;
    sh.grid_base = (int64_t)((sh.n_pairs + per_round - 1) / per_round) * 8 * sh.slots_per_pair;
}

// Workgroup shape and key-range split of a launch.  The key range is split only when the
// launch would leave most CUs without a workgroup (one long sequence prefilled in chunks)
// and every split still gets a few tiles.
static void prefill_grid(PrefillShape& sh, int B, int H_q, int H_kv, int max_seqlen_q, int rows_per_wave) {
    const int G = H_q / H_kv;
    sh.nw = 4;
    sh.hpw = gcd_int(G, sh.nw);
    sh.rows_per_wg = rows_per_wave * (sh.nw / sh.hpw);
    sh.n_qtiles = (max_seqlen_q + sh.rows_per_wg - 1) / sh.rows_per_wg;
    sh.wg_per_pair = sh.n_qtiles * (G / sh.hpw);
    sh.n_pairs = B * H_kv;
    prefill_set_grid(sh);
}

static PrefillShape prefill_shape(int B, int H_q, int H_kv, int max_seqlen_q, int max_seqlen_k) {
    PrefillShape sh;
    const int tiles = (max_seqlen_k + kBN - 1) / kBN;
    const nv_tuning tn = tuning();
    // The hand-scheduled kernel runs one workgroup per CU and never splits the key range: it takes the
    // launches that fill the chip with 64-row waves AND have long key ranges (nv_tuning.prefill_kernel
    // forces either kernel).  Its fixed cost per workgroup is about 12 us against 8 us here, its cost per
    // key tile lower: measured with 4096 workgroups it is 36 % slower at 1 tile per workgroup, 15 % at 4,
    // equal at 8 and 9 - 13 % faster from 16 tiles on; on whole-prompt prefills (several query tiles per
    // workgroup) it is 5 % faster at 512 tokens (8 tiles), 13 % at 1024, 18 % at 2048.
    prefill_grid(sh, B, H_q, H_kv, max_seqlen_q, 64);
    const int64_t big_wgs = (int64_t)sh.n_pairs * sh.wg_per_pair;
    // with fewer workgroups than CUs it splits the key range like the other kernel does (one workgroup per
    // CU, at least 4 tiles per split) and leaves unnormalised states for its merge kernel.  Measured on one
    // 512-query chunk (64 workgroups), launch pair, this kernel with four key ranges against the other kernel
    // unsplit: 1024 keys 22.6 against 24.3 us, 1536 keys 25.9 against 34.7, 2048 keys 28.8 against 36.4; at 512
    // keys (8 tiles) the other kernel wins, 14.7 against 19.5 us.
    int ns_asm = 1;
    if (tn.prefill_split > 0) {
        ns_asm = tn.prefill_split;
    } else if (big_wgs < 256) {
        ns_asm = (int)(256 / big_wgs);
        if (ns_asm > tiles / 4) ns_asm = tiles / 4;
        if (ns_asm < 1) ns_asm = 1;
    }
    const bool asm_fills_chip = big_wgs >= 256 || (ns_asm >= 2 && big_wgs * ns_asm >= 128);
    sh.hand_scheduled = tn.prefill_kernel >= 0 ? tn.prefill_kernel == 1
                                               : (asm_fills_chip && tiles >= (big_wgs >= 256 ? 8 : 16));
    if (tiles > prefill_asm_max_tiles()) sh.hand_scheduled = false;
    sh.items_per_wg = 1;
    if (sh.hand_scheduled) {
        if (ns_asm > tiles) ns_asm = tiles > 0 ? tiles : 1;
        sh.tiles_per_split = tiles > 0 ? (tiles + ns_asm - 1) / ns_asm : 1;
        sh.nsplit = ns_asm > 1 ? (tiles + sh.tiles_per_split - 1) / sh.tiles_per_split : 1;
        // several query tiles per workgroup (equal work for all workgroups of a sequence, a half or a quarter
        // of the fixed per-workgroup cost: about 6 us until the first MFMA) while that still leaves two
        // workgroups per CU
        if (sh.nsplit == 1) {
            const int groups = sh.wg_per_pair / sh.n_qtiles;
            for (int m = kMaxItemsPerWg; m >= 2; m /= 2) {
                const int slots = (sh.n_qtiles + m - 1) / m;
                if ((int64_t)sh.n_pairs * slots * groups >= 512) {
                    sh.items_per_wg = m;
                    sh.wg_per_pair = slots * groups;
                    sh.n_qtiles = slots;
                    prefill_set_grid(sh);
                    break;
                }
            }
        }
        return sh;
    }
    prefill_grid(sh, B, H_q, H_kv, max_seqlen_q, 32);
    int ns = 1;
    if (tn.prefill_split > 0) {
        ns = tn.prefill_split;
    } else {
        const int64_t wgs = (int64_t)sh.n_pairs * sh.wg_per_pair;
        const int64_t target = 512;  // two workgroups per CU
        if (wgs * 2 <= target) ns = (int)(target / wgs);
        const int max_by_len = tiles / 16;  // at least 16 tiles (1024 keys) per split
        if (ns > max_by_len) ns = max_by_len;
    }
    if (ns > tiles) ns = tiles;
    if (ns < 1) ns = 1;
    sh.tiles_per_split = (tiles + ns - 1) / ns;
    if (sh.tiles_per_split < 1) sh.tiles_per_split = 1;
    sh.nsplit = ns > 1 ? (tiles + sh.tiles_per_split - 1) / sh.tiles_per_split : 1;
    return sh;
}

}  // namespace nv

extern "C" int64_t nv_attn_prefill_workspace_bytes(int B, int total_q, int max_seqlen_q,
                                                  int max_seqlen_k, int H_q, int H_kv, int D) {
    using namespace nv;
    if (B <= 0 || total_q <= 0 || max_seqlen_q <= 0 || max_seqlen_k < 0 || H_q <= 0 || H_kv <= 0 ||
        H_q % H_kv != 0 || D != kHeadDim)
        return 0;
    const PrefillShape sh = prefill_shape(B, H_q, H_kv, max_seqlen_q, max_seqlen_k);
    if (sh.nsplit == 1) return 16;
    if (sh.hand_scheduled) return prefill_asm_state_bytes(sh.grid_base, sh.nsplit);
    return (int64_t)total_q * H_q * sh.nsplit * kPartStride * (int64_t)sizeof(float);
}

extern "C" int nv_attn_prefill_varlen(const void* q, const void* k_cache, const void* v_cache,
                                      void* o, const int32_t* cu_seqlens_q,
                                      const int32_t* cu_seqlens_k, const int32_t* block_tables,
                                      int B, int max_blocks, int max_seqlen_q, int max_seqlen_k,
                                      int H_q, int H_kv, int D, int64_t num_blocks, int block_size,
                                      float scale, int64_t q_stride_t, int64_t o_stride_t,
                                      int skip_single_query, int total_q, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
    using namespace nv;
    NV_REQUIRE(B >= 0 && max_blocks > 0 && max_seqlen_q >= 0 && max_seqlen_k >= 0 && H_q > 0 &&
                   H_kv > 0 && num_blocks > 0 && block_size > 0 && total_q >= 0,
               NV_ERR_INVALID_ARG, "nv_attn_prefill_varlen: bad sizes B=%d max_q=%d max_k=%d", B,
               max_seqlen_q, max_seqlen_k);
    NV_REQUIRE(D == kHeadDim, NV_ERR_UNSUPPORTED, "nv_attn_prefill_varlen: head_dim %d (128 only)", D);
    NV_REQUIRE(H_q % H_kv == 0, NV_ERR_INVALID_ARG, "nv_attn_prefill_varlen: H_q %% H_kv != 0");
    NV_REQUIRE(block_size % kBN == 0, NV_ERR_UNSUPPORTED,
               "nv_attn_prefill_varlen: block_size %d must be a multiple of %d", block_size, kBN);
    if (B == 0 || max_seqlen_q == 0 || total_q == 0) return NV_OK;
    NV_REQUIRE(q && k_cache && v_cache && o && cu_seqlens_q && cu_seqlens_k && block_tables,
               NV_ERR_INVALID_ARG, "nv_attn_prefill_varlen: null pointer");
    NV_REQUIRE(q_stride_t % 8 == 0 && o_stride_t % 8 == 0 && q_stride_t >= (int64_t)H_q * D &&
                   o_stride_t >= (int64_t)H_q * D,
               NV_ERR_INVALID_ARG, "nv_attn_prefill_varlen: bad token strides");
    NV_REQUIRE(((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)o |
                (uintptr_t)workspace) % 16 == 0,
               NV_ERR_INVALID_ARG, "nv_attn_prefill_varlen: pointers must be 16-byte aligned");
    NV_REQUIRE((int64_t)max_blocks * block_size >= max_seqlen_k, NV_ERR_INVALID_ARG,
               "nv_attn_prefill_varlen: block table too narrow for max_seqlen_k");

    const PrefillShape sh = prefill_shape(B, H_q, H_kv, max_seqlen_q, max_seqlen_k);
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
    p.hpw = sh.hpw;
    p.rows_per_wg = sh.rows_per_wg;
    p.n_qtiles = sh.n_qtiles;
    p.wg_per_pair = sh.wg_per_pair;
    p.n_pairs = sh.n_pairs;
    p.num_blocks = num_blocks;
    p.q_stride_vec = q_stride_t / 8;
    p.o_stride = o_stride_t;
    p.scale_log2 = scale * 1.4426950408889634f;
    p.skip_single = skip_single_query;
    p.part = (float*)workspace;
    p.nsplit = sh.nsplit;
    p.tiles_per_split = sh.tiles_per_split;
    p.grid_base = (int)sh.grid_base;
    p.xs_shift = sh.xs_shift;
    p.slots_per_pair = sh.slots_per_pair;
    p.fd_slots_per_pair = make_fastdiv((uint32_t)sh.slots_per_pair);
    p.pair_items = sh.items_per_wg;
    p.groups = sh.wg_per_pair / sh.n_qtiles;
    p.fd_wg_per_pair = make_fastdiv((uint32_t)sh.wg_per_pair);
    p.fd_groups = make_fastdiv((uint32_t)p.groups);
    p.fd_H_kv = make_fastdiv((uint32_t)H_kv);
    p.fd_hpw = make_fastdiv((uint32_t)sh.hpw);
    p.fd_rows_per_wg = make_fastdiv((uint32_t)sh.rows_per_wg);
    if (sh.nsplit > 1) {
        const int64_t need = sh.hand_scheduled
                                 ? prefill_asm_state_bytes(sh.grid_base, sh.nsplit)
                                 : (int64_t)total_q * H_q * sh.nsplit * kPartStride * (int64_t)sizeof(float);
        NV_REQUIRE(workspace && workspace_bytes >= need, NV_ERR_WORKSPACE,
                   "nv_attn_prefill_varlen: workspace %lld bytes < required %lld",
                   (long long)workspace_bytes, (long long)need);
    }
    const int64_t grid = sh.grid_base * sh.nsplit;
    NV_REQUIRE(grid < (1LL << 31) && (sh.nsplit == 1 || sh.grid_base < (1LL << 27)), NV_ERR_UNSUPPORTED,
               "nv_attn_prefill_varlen: launch too large");   // (the merge of a split launch has 8 workgroups per workgroup)
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (sh.hand_scheduled) {
        launch_prefill_asm(p, grid, st);
        rc = check_launch("nv_attn_prefill_varlen(hand-scheduled)");
    } else {
        hipLaunchKernelGGL(attn_prefill_kernel, dim3((unsigned)grid), dim3(256), 0, st, p);
        rc = check_launch("nv_attn_prefill_varlen");
    }
    if (rc != NV_OK || sh.nsplit == 1) return rc;
    if (sh.hand_scheduled) {
        launch_prefill_asm_merge(p, st);
        return check_launch("nv_attn_prefill_varlen(merge)");
    }
    hipLaunchKernelGGL(attn_prefill_merge_kernel,
                       dim3((unsigned)(((int64_t)max_seqlen_q * H_q + 7) / 8), (unsigned)B),
                       dim3(256), 0, st, p);
    return check_launch("nv_attn_prefill_varlen(merge)");
}
