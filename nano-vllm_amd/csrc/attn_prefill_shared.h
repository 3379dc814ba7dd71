// Declarations shared by the two prefill attention kernels (attn_prefill.hip, attn_prefill_asm.hip).
#pragma once

#include "nv_common.h"

namespace nv {

constexpr int kBN = 64;                     // keys per tile
constexpr float kRunInit = -1.0e30f;        // finite initial running max (keeps exp2 NaN-free)
constexpr float kRescaleThreshold = 8.0f;   // log2 units: P stays below 2^8 between rescales

constexpr int kPartStride = kHeadDim + 4;    // fp32 per partial state: 128 acc + m + l + 2 pad
// hand-scheduled kernel: one unit per (wave, key range) = its 64 rows in register order
// [qb 2][d 4][half 2][lane 64][8 bf16] followed by [qb 2][lane 64][m, l] (fp32)
constexpr int kImageAcc = 2 * 4 * 2 * 64 * 16;
constexpr int kImageUnit = kImageAcc + 2 * 64 * 8;

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
    int B, max_blocks, H_q, H_kv, block_size;
    int hpw;          // query heads per workgroup (divides G and 4)
    int rows_per_wg;  // 32 * (4 / hpw)
    int n_qtiles;     // ceil(max_seqlen_q / rows_per_wg)
    int wg_per_pair;  // n_qtiles * (G / hpw): workgroups of one (sequence, kv head)
    int n_pairs;      // B * H_kv
    int64_t num_blocks;
    int64_t q_stride_vec, o_stride;
    float scale_log2;
    int skip_single;
    // key-range splitting (launches with too few workgroups for the chip)
    float* part;          // [total_q, H_q, nsplit, kPartStride] fp32 partial states (hand-scheduled kernel:
                          // [grid_base * 4 waves, nsplit] units of kImageUnit bytes, accumulators in bf16)
    int nsplit;           // 1: results are written directly
    int tiles_per_split;  // 64-key tiles per split
    int grid_base;        // workgroups per split (multiple of 8)
    // A pair's workgroups sit on one XCD (they share its K/V tiles through that L2) unless the launch has fewer than 8
    // pairs: then each pair is spread over 1 << xs_shift XCDs (2, 4 or 8), or most of the chip would idle
    // (one sequence on a one-kv-head shard ran on one XCD of eight).  slots_per_pair = ceil(wg_per_pair >> xs_shift).
    int xs_shift, slots_per_pair;
    int pair_items;       // hand-scheduled kernel: query tiles per workgroup (1, 2 or 4; balanced over the sequence)
    // the block-index decomposition's divisors, prepared by the host (the merge launch is a few hundred instructions long)
    int groups;           // wg_per_pair / n_qtiles: head groups per kv head (G / hpw)
    FastDiv fd_wg_per_pair, fd_slots_per_pair, fd_groups, fd_H_kv, fd_hpw, fd_rows_per_wg;
};

// hand-scheduled kernel for large launches (attn_prefill_asm.hip): one wave per SIMD, 64 query rows per wave
void launch_prefill_asm(const PrefillParams& p, int64_t grid, hipStream_t st);
void launch_prefill_asm_merge(const PrefillParams& p, hipStream_t st);   // combines its key-range states
int64_t prefill_asm_state_bytes(int64_t grid_base, int nsplit);
int prefill_asm_max_tiles();

}  // namespace nv
