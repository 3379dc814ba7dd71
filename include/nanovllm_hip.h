/*
 * nanovllm_hip.h — C ABI of the gfx950 (MI355X) attention hot path.
 *
 * Every entry point
 *   - takes raw DEVICE pointers, plain sizes/strides and a HIP stream (passed as void*,
 *     i.e. a hipStream_t; NULL = the default stream),
 *   - only enqueues work on that stream: it never synchronises and, unless stated
 *     otherwise, never allocates,
 *   - returns NV_OK (0) or a negative NV_ERR_* code; nv_strerror() names the code and
 *     nv_last_error() gives the detail message of the calling thread's last failure.
 *
 * Data types are fixed: activations and KV cache are bf16 (passed as void*), metadata is
 * int32 except `positions` (int64), the rotary table is fp32.  head_dim must be 128.
 *
 * The reference (pradsgit/nano-vllm) has no FFI layer: the functions below replace the
 * device work its three layer modules hand to flash-attn / ATen / Inductor.  Each
 * declaration cites the reference lines it stands in for (paths relative to the
 * reference checkout).
 */
#ifndef NANOVLLM_HIP_H
#define NANOVLLM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* exported symbol marker (the library is built with hidden default visibility) */
#define NV_API __attribute__((visibility("default")))

#define NV_OK 0
#define NV_ERR_INVALID_ARG (-1)   /* null pointer, negative size, unsupported head_dim ... */
#define NV_ERR_UNSUPPORTED (-2)   /* a shape the kernels are not built for               */
#define NV_ERR_WORKSPACE (-3)     /* workspace missing or smaller than required           */
#define NV_ERR_LAUNCH (-4)        /* hipLaunchKernel / hipGetLastError reported a failure */

NV_API const char* nv_strerror(int code);
NV_API const char* nv_last_error(void);
/* ABI version of this library (bumped on any signature change). */
NV_API int nv_abi_version(void);

/*
 * Launch-plan overrides (measurement and test aid; no counterpart in the reference, whose
 * flash-attn dependency picks its own splits).  The library reads NO environment variables: the
 * only way to steer kernel selection is this process-wide setting.  Every field's "heuristic"
 * value (the value after nv_set_tuning(NULL)) lets the library choose.  nv_attn_*_workspace_bytes
 * and the matching launch compute their plan from the same setting, so it must not change between
 * the two calls (a launch whose plan outgrew the workspace answers NV_ERR_WORKSPACE, never
 * overruns it).  Not meant to be changed while launches are being issued from other threads.
 */
typedef struct nv_tuning {
    int decode_target_waves;  /* 0 = heuristic; > 0: split keys until a launch has about this many waves */
    int decode_chunk;         /* 0 = heuristic; > 0 (multiple of 64): keys per decode work item          */
    int decode_work_item;     /* -1 = heuristic; 0 = one wave per work item; 1 = one workgroup            */
    int decode_score_path;    /* -1 = heuristic; 0 = VALU scores for H_q/H_kv = 4, 8; 1 = MFMA scores     */
    int decode_mfma_depth;    /* -1 = heuristic; 0 / 1 = tiles prefetched ahead in the MFMA-score kernel  */
    int decode_single_launch; /* -1, 0 = split sequences combined by a second launch; 1 = inside the launch */
    int prefill_split;        /* 0 = heuristic; > 0: forced number of key-range splits                    */
    int prefill_kernel;       /* -1 = heuristic; 0 = compiler-scheduled kernel; 1 = hand-scheduled kernel  */
} nv_tuning;
NV_API int nv_set_tuning(const nv_tuning* t);   /* NULL restores the heuristics */
NV_API int nv_get_tuning(nv_tuning* out);

/*
 * KV-cache scatter.  Replaces Attention.store_kvcache, nanovllm/layers/attention.py:210-219:
 *   k_cache[slot // block_size, slot % block_size] = k[t]   (same for v)
 * k, v: [T, H_kv, D] bf16, token stride given in ELEMENTS (rows themselves contiguous).
 * k_cache, v_cache: [num_blocks, block_size, H_kv, D] bf16 contiguous.
 * slot_mapping: [T] int32.  Slots < 0 or >= num_blocks*block_size are skipped (the
 * reference never produces them on its live path; skipping keeps padding harmless).
 * Bit-exact copy.
 */
NV_API int nv_store_kvcache(const void* k, const void* v, void* k_cache, void* v_cache,
                     const int32_t* slot_mapping, int T, int H_kv, int D,
                     int64_t num_blocks, int block_size,
                     int64_t k_stride_t, int64_t v_stride_t, void* stream);

/*
 * Paged decode attention (every selected sequence has exactly one query token).
 * Replaces the flash_attn_varlen_func call at nanovllm/layers/attention.py:241-252 for
 * steps whose sequences all have seqlen_q == 1.
 *   q: [T, H_q, D] bf16 (token stride q_stride_t elements); o: [T, H_q, D] bf16
 *   (token stride o_stride_t elements).
 *   cu_seqlens_k: [B+1] int32; sequence b attends to keys 0 .. cu_k[b+1]-cu_k[b]-1, key j
 *   living at cache[block_tables[b, j / block_size], j % block_size, h_q / (H_q/H_kv)].
 *   cu_seqlens_q: [B+1] int32 or NULL.  NULL means token b belongs to sequence b.  When
 *   given, sequences with seqlen_q != 1 are skipped (they belong to the prefill kernel).
 *   block_tables: [B, max_blocks] int32, padded with -1 (padding is never dereferenced).
 *   max_seqlen_k: host-side upper bound of the per-sequence key count.
 *   workspace: device scratch of at least nv_attn_decode_workspace_bytes(...) bytes,
 *   16-byte aligned; contents need no initialisation.
 * Sequences with zero keys produce zeros.
 * Shapes: H_q a multiple of H_kv with 1 <= H_q / H_kv <= 8 (NV_ERR_UNSUPPORTED beyond);
 * any block_size >= 1 (multiples of 16 resolve the block once per 16 keys);
 * max_blocks * block_size >= max_seqlen_k.
 */
NV_API int64_t nv_attn_decode_workspace_bytes(int B, int H_q, int H_kv, int D, int max_seqlen_k);
NV_API int nv_attn_decode_paged(const void* q, const void* k_cache, const void* v_cache, void* o,
                         const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                         const int32_t* block_tables, int B, int max_blocks,
                         int max_seqlen_k, int H_q, int H_kv, int D,
                         int64_t num_blocks, int block_size, float scale,
                         int64_t q_stride_t, int64_t o_stride_t,
                         void* workspace, int64_t workspace_bytes, void* stream);

/*
 * The decode launch for hosts that know more of the batch than its longest sequence.  No counterpart in the
 * reference, whose flash_attn_varlen_func call (nanovllm/layers/attention.py:241-252) passes max_seqlen_k only --
 * but its runner holds every length as a host integer (nanovllm/engine/model_runner.py:85-89).
 * The launch plan cuts long sequences into key ranges of `chunk` keys, one work item each.  From max_seqlen_k
 * alone it cannot tell 256 sequences of 2048 keys (no cut wanted) from 255 of 640 keys and one of 8192 (whose
 * few waves end the launch long after the others unless its keys are spread over many work items).
 *   nv_attn_decode_suggest_chunk: keys per work item for a batch with this sum of lengths, or 0 when the
 *     plan's own choice stands: balanced batches (longest sequence within 1.4 x the mean length, or under 512
 *     keys beyond it; within 2.5 x for launches with few (sequence, kv head) pairs), and launches with more
 *     work items than are resident at once unless one sequence alone outlasts the rest (from about half of
 *     sum x H_kv / 256 keys on).  Such launches gain from rows handed over longest first instead: a launch
 *     takes its work in row order and ends with its last workgroup.  When their lengths are spread out (longest
 *     from 1.7 x the mean and 1024 keys on) the answer is 2048, the plan's own cap: it cuts nothing there, and
 *     any chunk_keys > 0 tells the plan that the lengths are uneven, for which it takes the prefetching kernel.
 *   nv_attn_decode_paged_chunked / nv_attn_decode_paged_fused_chunked: the entry points above with that value
 *     (chunk_keys; 0 = the plan's choice; rounded up to a multiple of 64 keys, at most 256 ranges per sequence)
 *     and flags (0, or NV_DECODE_WS_COUNTERS_ZERO below).
 *     Results equal the unhinted launch's up to the fp32 summation order of the merge.
 *   nv_attn_decode_workspace_bytes_chunked: the workspace such a launch needs.
 * Sequences of at most chunk keys are written by the attention launch directly, whatever the others need.
 */
/* flags of the _chunked entry points.  NV_DECODE_WS_COUNTERS_ZERO: the caller vouches that the first 64 KiB of the
 * workspace (the arrival counters of the in-launch combine) are zero and that nothing but this library's decode
 * launches, ordered on one stream, writes the workspace: launches with at most 8 (sequence, kv head) pairs then
 * combine their key ranges inside the launch, with no second kernel (the launch puts the counters back to zero).
 * One sequence of 1024 keys at 16 / 8 heads: 9.3 us against 14.4 with the merge launch. */
#define NV_DECODE_WS_COUNTERS_ZERO 1
NV_API int nv_attn_decode_suggest_chunk(int B, int H_q, int H_kv, int max_seqlen_k, int64_t sum_seqlen_k);
NV_API int64_t nv_attn_decode_workspace_bytes_chunked(int B, int H_q, int H_kv, int D, int max_seqlen_k,
                                                      int chunk_keys);
NV_API int nv_attn_decode_paged_chunked(const void* q, const void* k_cache, const void* v_cache, void* o,
                                        const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                        const int32_t* block_tables, int B, int max_blocks,
                                        int max_seqlen_k, int H_q, int H_kv, int D,
                                        int64_t num_blocks, int block_size, float scale,
                                        int64_t q_stride_t, int64_t o_stride_t,
                                        void* workspace, int64_t workspace_bytes, int chunk_keys, int flags,
                                        void* stream);

/*
 * nv_attn_decode_paged without a workspace argument, for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  Same arguments, kernels and results as nv_attn_decode_paged; the
 * split-key partial states go to an array built into the library's code object (32 MiB of device
 * memory per GPU the library runs on, held until the process ends).  The call allocates nothing.
 * A launch whose partial states need more than nv_attn_decode_static_scratch_bytes() is refused with
 * NV_ERR_WORKSPACE (use nv_attn_decode_paged with a caller workspace then).  Every call on a GPU
 * shares that GPU's array: calls that overlap in time on one GPU (different streams) race; calls on
 * one stream are ordered and safe.  nv_attn_decode_static_scratch_bytes() returns the capacity and
 * resolves the array's address on the current device (-1 on failure); calling it once per device
 * before a graph capture keeps that lookup out of the capture.  The array belongs to the CURRENT device:
 * a stream of another device is refused with NV_ERR_INVALID_ARG.  Launches with at most 8 (sequence, kv
 * head) pairs combine their key chunks inside the launch here (no second kernel), since the array's
 * arrival counters are known to be zero; results then agree with nv_attn_decode_paged's to fp32
 * summation order.
 */
NV_API int64_t nv_attn_decode_static_scratch_bytes(void);
NV_API int nv_attn_decode_paged_static(const void* q, const void* k_cache, const void* v_cache, void* o,
                                const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                const int32_t* block_tables, int B, int max_blocks,
                                int max_seqlen_k, int H_q, int H_kv, int D,
                                int64_t num_blocks, int block_size, float scale,
                                int64_t q_stride_t, int64_t o_stride_t, void* stream);

/*
 * Varlen prefill / chunked-prefill attention over the paged cache, causal mask aligned
 * to the bottom-right corner: query i of n_q sees keys j <= n_k - n_q + i.
 * Replaces the same call (nanovllm/layers/attention.py:241-252) for steps with any
 * seqlen_q > 1; argument wiring follows nanovllm/engine/model_runner.py:85-119.
 *   cu_seqlens_q, cu_seqlens_k: [B+1] int32.  max_seqlen_q: host-side upper bound.
 *   skip_single_query != 0 leaves sequences with seqlen_q == 1 untouched (the caller
 *   runs nv_attn_decode_paged for them); 0 processes every sequence here.
 *   total_q: number of query tokens (= cu_seqlens_q[B]); the block dimension y of the
 *   merge launch is B, so B <= 65535 when the key range is split.
 *   workspace: device scratch of at least nv_attn_prefill_workspace_bytes(...) bytes,
 *   16-byte aligned, contents need no initialisation.  It is used only when the launch
 *   would have too few workgroups for the chip (few long sequences): the key range is then
 *   split over several workgroups and a second launch combines their states.
 * Rows with no visible key produce zeros.
 * Shapes: H_q a multiple of H_kv (any ratio); block_size a multiple of 64
 * (NV_ERR_UNSUPPORTED otherwise); max_blocks * block_size >= max_seqlen_k.
 */
NV_API int64_t nv_attn_prefill_workspace_bytes(int B, int total_q, int max_seqlen_q, int max_seqlen_k,
                                        int H_q, int H_kv, int D);
NV_API int nv_attn_prefill_varlen(const void* q, const void* k_cache, const void* v_cache, void* o,
                           const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                           const int32_t* block_tables, int B, int max_blocks,
                           int max_seqlen_q, int max_seqlen_k, int H_q, int H_kv, int D,
                           int64_t num_blocks, int block_size, float scale,
                           int64_t q_stride_t, int64_t o_stride_t,
                           int skip_single_query, int total_q,
                           void* workspace, int64_t workspace_bytes, void* stream);

/*
 * RMSNorm.  Replaces RMSNorm.forward without residual, nanovllm/layers/layernorm.py:79-100:
 *   y = bf16( f32(x) * rsqrt(mean(f32(x)^2) + eps) * f32(w) )
 * x: [rows, cols] bf16 with row stride x_stride elements; y: [rows, cols] contiguous.
 * cols must be a multiple of 8 and at most 16384.  Also used per head (cols = head_dim).
 */
NV_API int nv_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int cols,
               int64_t x_stride, float eps, void* stream);

/*
 * SiluAndMul.  Replaces SiluAndMul.forward, nanovllm/layers/activation.py:10-16, as its
 * compiled form evaluates it:  y[r, j] = bf16( silu(f32(x[r, j])) * f32(x[r, inter + j]) ).
 * x: [rows, 2*inter] bf16 with row stride x_stride elements; y: [rows, inter] contiguous.
 * inter must be a multiple of 8.
 */
NV_API int nv_silu_and_mul(const void* x, void* y, int64_t rows, int inter, int64_t x_stride,
                    void* stream);

/*
 * Fused residual add + RMSNorm.  Replaces RMSNorm.forward with residual,
 * nanovllm/layers/layernorm.py:88-100:
 *   s = f32(x) + f32(res);  new_res = bf16(s);  y = bf16( s * rsqrt(mean(s^2)+eps) * f32(w) )
 * (y is computed from the unrounded s).  All tensors [rows, cols] contiguous bf16.
 * y may alias x and new_res may alias res.
 */
NV_API int nv_add_rmsnorm(const void* x, const void* res, const void* w, void* y, void* new_res,
                   int64_t rows, int cols, float eps, void* stream);

/*
 * Rotary embedding (NeoX rotate-half), replaces RotaryPositionEmbedding.forward,
 * nanovllm/layers/rope.py:45-56 with apply_rotary_emb :5-13:
 *   cs = cos_sin[pos];  y = [x1*cos - x2*sin, x1*sin + x2*cos]  in fp32, cast to bf16.
 * q: [T, H_q, D], k: [T, H_kv, D] bf16 (token strides in elements, heads contiguous);
 * q_out, k_out contiguous.  cos_sin: [max_pos, D] fp32 (cos | sin).  positions: [T] int64.
 * Positions outside [0, max_pos) are clamped into range.
 */
NV_API int nv_rope(const void* q, const void* k, void* q_out, void* k_out, const float* cos_sin,
            const int64_t* positions, int T, int H_q, int H_kv, int D, int64_t max_pos,
            int64_t q_stride_t, int64_t k_stride_t, void* stream);

/*
 * Fused per-head RMSNorm (q_norm / k_norm) + rotary embedding + KV-cache scatter: the
 * chain nanovllm/model/qwen3.py:67-75 (q_norm, k_norm, rope) followed by the store at
 * nanovllm/layers/attention.py:230-231, in one launch.
 *   q_w, k_w: [D] bf16 norm weights, or NULL for models without q/k-norm.
 *   The normalised value is rounded to bf16 before the rotation, as two separate
 *   modules would do.
 *   slot_mapping / k_cache / v_cache may all be NULL to skip the scatter; q_out / k_out
 *   receive the rotated tensors ([T,H,D] contiguous; k_out may be NULL when the scatter
 *   is requested and the caller does not need k).
 */
NV_API int nv_qknorm_rope_store(const void* q, const void* k, const void* v, const void* q_w,
                         const void* k_w, const float* cos_sin, const int64_t* positions,
                         const int32_t* slot_mapping, void* k_cache, void* v_cache,
                         void* q_out, void* k_out, int T, int H_q, int H_kv, int D,
                         int64_t max_pos, float eps, int64_t num_blocks, int block_size,
                         int64_t q_stride_t, int64_t k_stride_t, int64_t v_stride_t,
                         void* stream);

/*
 * A pure decode step's whole attention block in ONE launch: the chain nanovllm/model/qwen3.py:67-75
 * (q_norm, k_norm, rope), the store at nanovllm/layers/attention.py:230-231 and the flash_attn_varlen_func
 * call at :241-252 for a step in which every sequence has exactly one new token.  Equal, bit for bit, to
 * nv_qknorm_rope_store followed by nv_attn_decode_paged on the same inputs (the per-head arithmetic is one
 * function, and the attention sees the new token's key and value as it would read them back from the cache).
 *   q, k, v: the step's RAW projections, [B, H_q, D] / [B, H_kv, D] bf16, token strides in elements (slices of
 *   one fused projection are fine); token b belongs to sequence b.
 *   q_w, k_w, cos_sin, positions, max_pos, eps: as for nv_qknorm_rope_store.
 *   slot_mapping: [B] int32, the cache slot of sequence b's new token, i.e. of its key number
 *   cu_seqlens_k[b+1] - cu_seqlens_k[b] - 1 (the reference's runner builds it that way,
 *   nanovllm/engine/model_runner.py:71-81); a slot outside the cache is not written, the attention still
 *   sees the new token.
 *   k_cache, v_cache: read for keys 0 .. n_k - 2 and WRITTEN at the new token's slot.
 *   The remaining arguments, the workspace and the results are nv_attn_decode_paged's.
 * Shapes: H_q / H_kv of 1 .. 8 (3 .. 8 on the MFMA score path, which is the default for them) and block sizes that
 * are multiples of 16; NV_ERR_UNSUPPORTED otherwise (run the two launches then).
 */
NV_API int nv_attn_decode_paged_fused(const void* q, const void* k, const void* v, const void* q_w,
                               const void* k_w, const float* cos_sin, const int64_t* positions,
                               const int32_t* slot_mapping, void* k_cache, void* v_cache, void* o,
                               const int32_t* cu_seqlens_k, const int32_t* block_tables, int B,
                               int max_blocks, int max_seqlen_k, int H_q, int H_kv, int D,
                               int64_t num_blocks, int block_size, int64_t max_pos, float eps, float scale,
                               int64_t q_stride_t, int64_t k_stride_t, int64_t v_stride_t,
                               int64_t o_stride_t, void* workspace, int64_t workspace_bytes, void* stream);
/* ... with the host's choice of keys per work item (see nv_attn_decode_suggest_chunk above) */
NV_API int nv_attn_decode_paged_fused_chunked(const void* q, const void* k, const void* v, const void* q_w,
                               const void* k_w, const float* cos_sin, const int64_t* positions,
                               const int32_t* slot_mapping, void* k_cache, void* v_cache, void* o,
                               const int32_t* cu_seqlens_k, const int32_t* block_tables, int B,
                               int max_blocks, int max_seqlen_k, int H_q, int H_kv, int D,
                               int64_t num_blocks, int block_size, int64_t max_pos, float eps, float scale,
                               int64_t q_stride_t, int64_t k_stride_t, int64_t v_stride_t,
                               int64_t o_stride_t, void* workspace, int64_t workspace_bytes, int chunk_keys,
                               int flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NANOVLLM_HIP_H */
