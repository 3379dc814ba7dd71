// Paged decode attention (one query token per sequence) for gfx950.
//
// HBM-bandwidth-bound: every K/V byte of the context is read exactly once per step, so the
// kernels are built around keeping many 16-byte-per-lane loads in flight.
//
// Two kernels share the work decomposition and the epilogue:
//   attn_decode_kernel       G = H_q / H_kv <= 2 (and a comparison path for 4, 8): all VALU.
//   attn_decode_mfma_kernel  G = 3 .. 8: the scores of a 16-key tile come from four MFMAs.
// Work item = (sequence b, kv head h, key chunk s), streamed in tiles of 16 keys.  It is one
// WAVE (four independent items per workgroup, neighbouring kv heads of the same tokens) or,
// for launches with few (sequence, head) pairs, one WORKGROUP whose four waves take a
// quarter of the chunk each and merge their states in LDS.  V (and K in the VALU kernel) is
// loaded with lane l holding 16 bytes (8 dims) of key (l >> 4) of each 4-key load, so one
// wave load instruction covers 4 whole 256-byte rows of the head.  Every 16-lane row keeps
// an online-softmax state (m, l, acc) for the G query heads of the kv head; rows, waves and
// chunks are merged at the end (registers, LDS, and a second small kernel respectively).
//
// Semantics: flash_attn_varlen_func(..., causal=True, block_table=...) with seqlen_q == 1
// as called at nanovllm/layers/attention.py:241-252 (the query sees every cached key,
// including the one stored earlier in the same forward, :230-231).
#include "nv_common.h"
#include "norm_rope_shared.h"
#include <type_traits>

// Build-time tuning knobs (defaults are the measured best on MI355X).
#ifndef NV_DEC_WAVES
#define NV_DEC_WAVES 4   // waves (work items) per workgroup
#endif
#ifndef NV_DEC_NT
#define NV_DEC_NT 1      // 1: K/V loads carry the non-temporal hint (read-once stream)
#endif

#ifndef NV_DEC_TARGET_WAVES
#define NV_DEC_TARGET_WAVES 2048   // split the keys until the launch has about this many waves
#endif

namespace nv {

constexpr int kDecWaves = NV_DEC_WAVES;
constexpr float kNegBig = -1.0e30f;  // finite "minus infinity": keeps exp2(m_old - m_new) NaN-free
constexpr int kPartStride = kHeadDim + 4;  // fp32 per partial: 128 acc + m + l + 2 pad (16-B rows)

__device__ __forceinline__ float dot8(const uint4& a, const uint4& b) {
    float d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.x),
                                              __builtin_bit_cast(bf16x2, b.x), 0.f, false);
    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.y),
                                        __builtin_bit_cast(bf16x2, b.y), d, false);
    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.z),
                                        __builtin_bit_cast(bf16x2, b.z), d, false);
    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.w),
                                        __builtin_bit_cast(bf16x2, b.w), d, false);
    return d;
}

__device__ __forceinline__ uint4 kv_load(const uint4* p) {
#if NV_DEC_NT
    return load_stream16(p);
#else
    return *p;
#endif
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// c ? a : b on values (the conditional operator on two uint4 lvalues selects between their ADDRESSES, which
// keeps both objects in memory instead of registers)
__device__ __forceinline__ uint4 pick(bool c, const uint4 a, const uint4 b) {
    return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}

// 16-byte write-through store (sc1): the bytes go to memory instead of staying dirty in this XCD's L2.
// The trailing s_nop keeps the compiler's next instruction off the data registers until the store has
// read them.  The compiler does not count this store: drain with an explicit s_waitcnt vmcnt(0).
__device__ __forceinline__ void store16_wt(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

struct DecodeParams {
    const uint4* q;
    const uint4* k_cache;
    const uint4* v_cache;
    uint4* o;
    const int32_t* cu_q;
    const int32_t* cu_k;
    const int32_t* block_tables;
    float* part;  // [B, H_q, nsplit, kPartStride]
    int* tickets; // [B, H_kv] arrival counters of the in-launch combine (zero before the launch), or null
    int B, max_blocks, H_kv, block_size;
    int chunk, nsplit;
    int group_items;  // work items per key-range group of the grid (B * H_kv, rounded up to whole workgroups)
    int group_rot;    // items by which each group is rotated against the one before it
    int G;  // query heads per kv head
    int64_t num_blocks;
    int64_t q_stride_vec, o_stride_vec;
    float scale_log2;
    // FUSE instantiations only (nv_attn_decode_paged_fused): q is the RAW query projection, and the step's new
    // key / value rows arrive as projections too; the kernel normalises and rotates them itself
    const uint4* k_new;       // [B, H_kv, D] raw key projection of the step's token
    const uint4* v_new;       // [B, H_kv, D] value projection
    const uint4* q_w;         // [D] q-norm weight, or null (no q/k-norm)
    const uint4* k_w;
    const float4* cos_sin;    // [max_pos, D] fp32: cos | sin
    const int64_t* positions; // [B]
    const int32_t* slots;     // [B] cache slot of the new token
    uint4* k_cache_w;         // the caches again, writable
    uint4* v_cache_w;
    int64_t k_stride_vec, v_stride_vec, max_pos, num_slots;
    float eps;
};

// Final state of one (sequence, query head, key chunk) held by 16 lanes (lane c: dims 8c..8c+7):
// written as the bf16 output when the sequence is not split, else as an fp32 partial state.
// `direct`: the sequence has a single key range in this launch (always when the launch is unsplit; in a split
// launch, every sequence no longer than one chunk): its output row is final, no partial state, no merge work.
template <bool TK>
__device__ __forceinline__ void store_state(const DecodeParams& p, int b, int hq, int split, int tq, int c,
                                            float m, float l, const float (&a)[8], bool direct) {
    if (direct) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        uint4 ov;
        ov.x = pack_bf16(a[0] * inv, a[1] * inv);
        ov.y = pack_bf16(a[2] * inv, a[3] * inv);
        ov.z = pack_bf16(a[4] * inv, a[5] * inv);
        ov.w = pack_bf16(a[6] * inv, a[7] * inv);
        p.o[(int64_t)tq * p.o_stride_vec + (int64_t)hq * 16 + c] = ov;
    } else {
        const int H_q = p.H_kv * p.G;
        float* dst = p.part + (((int64_t)b * H_q + hq) * p.nsplit + split) * kPartStride;
        if (TK) {
            // in-launch combine: another workgroup reads this state before the launch ends, so the stores
            // are write-through (they leave the XCD's L2 for memory; the storing wave drains them before
            // it signals) and every 16-byte store covers whole elements of the state
            store16_wt(dst + 8 * c, f32x4{a[0], a[1], a[2], a[3]});
            store16_wt(dst + 8 * c + 4, f32x4{a[4], a[5], a[6], a[7]});
            if (c == 0) store16_wt(dst + kHeadDim, f32x4{m, l, 0.f, 0.f});
        } else {
            float4* d4 = reinterpret_cast<float4*>(dst + 8 * c);
            d4[0] = make_float4(a[0], a[1], a[2], a[3]);
            d4[1] = make_float4(a[4], a[5], a[6], a[7]);
            if (c == 0) {
                dst[kHeadDim] = m;
                dst[kHeadDim + 1] = l;
            }
        }
    }
}

// Workgroup-split form: the kDecWaves waves of a workgroup each streamed a quarter of one
// (sequence, kv head, chunk); their states meet in LDS and wave w finishes heads w, w+4, ...
// sm: [kDecWaves][G][kPartStride] floats.  Every wave of the workgroup must call this.
template <int G, bool TK>
__device__ __forceinline__ void wg_merge_store(const DecodeParams& p, float* sm, int wave, int lane, int b, int h,
                                               int split, int tq, const float (&m)[G], const float (&l)[G],
                                               const float (&acc)[G][8], bool direct) {
    const int c = lane & 15;
    if (lane < 16) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float* dst = sm + (wave * G + g) * kPartStride;
            float4* d4 = reinterpret_cast<float4*>(dst + 8 * c);
            d4[0] = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
            d4[1] = make_float4(acc[g][4], acc[g][5], acc[g][6], acc[g][7]);
            if (c == 0) {
                dst[kHeadDim] = m[g];
                dst[kHeadDim + 1] = l[g];
            }
        }
    }
    __syncthreads();
    if (lane >= 16) return;
    for (int g = wave; g < p.G; g += kDecWaves) {   // p.G <= G: the template may be padded
        float mm = kNegBig;
#pragma unroll
        for (int w = 0; w < kDecWaves; ++w) mm = fmaxf(mm, sm[(w * G + g) * kPartStride + kHeadDim]);
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float ls = 0.f;
#pragma unroll
        for (int w = 0; w < kDecWaves; ++w) {
            const float* src = sm + (w * G + g) * kPartStride;
            const float f = __builtin_amdgcn_exp2f(src[kHeadDim] - mm);
            const float4 x = *reinterpret_cast<const float4*>(src + 8 * c);
            const float4 y = *reinterpret_cast<const float4*>(src + 8 * c + 4);
            a[0] += x.x * f; a[1] += x.y * f; a[2] += x.z * f; a[3] += x.w * f;
            a[4] += y.x * f; a[5] += y.y * f; a[6] += y.z * f; a[7] += y.w * f;
            ls += src[kHeadDim + 1] * f;
        }
        store_state<TK>(p, b, h * p.G + g, split, tq, c, mm, ls, a, direct);
    }
}

// ---- in-launch combine of split sequences --------------------------------------------------------
// With p.tickets set, the work item that stores the LAST partial state of a (sequence, kv head) also
// combines all of them and writes the output rows, so a split launch needs no second kernel.
// Hand-off between workgroups (any placement over the XCDs): write-through partial stores -> every
// storing wave drains its stores -> (workgroup items: barrier) -> one lane: relaxed agent-scope add on the
// pair's counter; the item that draws ns - 1 runs one agent-scope acquire and then reads every partial
// with plain vector loads.  That item puts the counter back to zero for the next launch.
// (Plain stores behind an agent-scope release in every item measured 1.5x slower for the whole launch:
// the release writes back the XCD's whole L2 while every other wave is streaming through it.)

// One 32-lane group walks partial states s0, s0 + step, ... of one (sequence, query head); lane d4 holds
// dims 4 d4 .. 4 d4 + 3.  Online form: one pass, every load independent of the others.
__device__ __forceinline__ void merge_walk(const float* base, int s0, int step, int ns, int d4, float& mm,
                                           float& den, float4& num) {
    mm = kNegBig;
    den = 0.f;
    num = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int s = s0; s < ns; s += step) {
        const float2 ml = *reinterpret_cast<const float2*>(base + s * kPartStride + kHeadDim);
        const float4 a = *reinterpret_cast<const float4*>(base + s * kPartStride + 4 * d4);
        const float mn = fmaxf(mm, ml.x);
        const float fo = __builtin_amdgcn_exp2f(mm - mn), fn = __builtin_amdgcn_exp2f(ml.x - mn);
        num.x = num.x * fo + a.x * fn;
        num.y = num.y * fo + a.y * fn;
        num.z = num.z * fo + a.z * fn;
        num.w = num.w * fo + a.w * fn;
        den = den * fo + ml.y * fn;
        mm = mn;
    }
}

__device__ __forceinline__ void merge_store_row(const DecodeParams& p, int tq, int hq, int d4, float den,
                                                const float4& num) {
    const float r = den > 0.f ? 1.0f / den : 0.f;
    uint2 w;
    w.x = pack_bf16(num.x * r, num.y * r);
    w.y = pack_bf16(num.z * r, num.w * r);
    __bf16* o = reinterpret_cast<__bf16*>(p.o) + (int64_t)tq * p.o_stride_vec * 8 + hq * kHeadDim;
    *reinterpret_cast<uint2*>(o + 4 * d4) = w;
}

// Arrival of one work item at its pair's counter; true for the item that has to combine.  `leader` is
// one lane of the item; every wave of the item that stored partial states has drained them already
// (and, for a workgroup item, passed a barrier after that).
__device__ __forceinline__ int ticket_arrive(int* ticket) {
    return __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void ticket_acquire_and_reset(int* ticket) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave combines the partial states of query heads g0, g0 + gstep, ... of kv head h: its two halves walk
// the even and the odd states, lane (lane & 31) holds 4 dims.
__device__ __forceinline__ void wave_merge_heads(const DecodeParams& p, int b, int h, int tq, int ns, int lane,
                                                 int g0, int gstep) {
    const int H_q = p.H_kv * p.G;
    const int grp = lane >> 5, d4 = lane & 31;
    for (int g = g0; g < p.G; g += gstep) {
        const int hq = h * p.G + g;
        const float* base = p.part + ((int64_t)b * H_q + hq) * p.nsplit * kPartStride;
        float mm, den;
        float4 num;
        merge_walk(base, grp, 2, ns, d4, mm, den, num);
        const float mo = __shfl_xor(mm, 32);
        const float mn = fmaxf(mm, mo);
        const float f = __builtin_amdgcn_exp2f(mm - mn);
        num.x *= f; num.y *= f; num.z *= f; num.w *= f;
        den *= f;
        num.x += __shfl_xor(num.x, 32);
        num.y += __shfl_xor(num.y, 32);
        num.z += __shfl_xor(num.z, 32);
        num.w += __shfl_xor(num.w, 32);
        den += __shfl_xor(den, 32);
        if (grp == 0) merge_store_row(p, tq, hq, d4, den, num);
    }
}

// Work item = one wave.  All 64 lanes call this after the wave's partial stores.
__device__ __forceinline__ void wave_ticket_merge(const DecodeParams& p, int b, int h, int tq, int n_k, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ns = (n_k + p.chunk - 1) / p.chunk;
    ns = ns < 1 ? 1 : (ns > p.nsplit ? p.nsplit : ns);
    int* ticket = p.tickets + (int64_t)b * p.H_kv + h;
    int old = 0;
    if (lane == 0) old = ticket_arrive(ticket);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != ns - 1) return;
    if (lane == 0) ticket_acquire_and_reset(ticket);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_merge_heads(p, b, h, tq, ns, lane, 0, 1);
}

// Work item = workgroup of kDecWaves waves.  Every thread calls this after wg_merge_store; flag is one
// LDS word.  The waves of the combining workgroup take the query heads in turn.
__device__ __forceinline__ void wg_ticket_merge(const DecodeParams& p, int* flag, int b, int h, int tq, int n_k,
                                                int wave, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int ns = (n_k + p.chunk - 1) / p.chunk;
    ns = ns < 1 ? 1 : (ns > p.nsplit ? p.nsplit : ns);
    int* ticket = p.tickets + (int64_t)b * p.H_kv + h;
    if (threadIdx.x == 0) *flag = ticket_arrive(ticket);
    __syncthreads();
    const int old = *flag;
    if (old != ns - 1) return;   // uniform over the workgroup
    if (threadIdx.x == 0) ticket_acquire_and_reset(ticket);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    wave_merge_heads(p, b, h, tq, ns, lane, wave, kDecWaves);
}

// kDepth = tiles (8 KiB per wave each) kept in flight ahead of the one being consumed.
// WGS = false: every wave is its own work item.  WGS = true (launches with few (sequence, head)
// pairs): the workgroup is one work item, its waves take a quarter of the chunk each and merge
// through LDS, which replaces a 4-fold deeper global split (partial states and merge work / 4).
//
// FUSE (pure decode steps, sequence b = token b): q/k-norm, rotary embedding and the KV-cache scatter of the
// step's token run in front of the key stream instead of as a launch of their own.  Every work item of a
// (sequence, kv head) normalises and rotates its G query heads from the raw projection (the arithmetic is
// head_norm_rope, the function the separate launch uses: same bits); the item whose keys include the new
// token's position n_k - 1 does the same for the new key, writes it and the value row to the cache slot and
// puts both into its registers in place of what the tile load fetched from that (still unwritten) slot.
template <int G, bool ALIGNED, int kDepth, bool WGS, bool TK, bool FUSE = false>
__global__ __launch_bounds__(kDecWaves * 64) void attn_decode_kernel(const DecodeParams p) {
    // workgroup items: the waves' states (its first word later carries the arrival number of the in-launch combine)
    __shared__ __attribute__((aligned(16))) float wg_sm[WGS ? kDecWaves * G * kPartStride : 4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane >> 4;  // which of the 4 keys of a load this lane holds
    const int c = lane & 15;   // which 16-byte piece of the 256-byte head row

    // item -> (b, split, h), h fastest: the waves of a workgroup read neighbouring head
    // rows of the same tokens, i.e. whole contiguous cache rows.
    const int item = WGS ? (int)blockIdx.x : blockIdx.x * kDecWaves + wave;
    // Key-range-major: the items of the batch's LAST key range first, ..., those of range 0 last; within a range
    // (sequence, kv head) with the head fastest.  A batch of short sequences with one long one has work in range 0
    // only, except for that sequence: sequence-major order would put the few non-empty workgroups at a fixed stride
    // in the grid -- always the same XCDs and CUs under round-robin dispatch -- with empty ones between; last
    // ranges first puts the longest sequences' extra work in flight before the bulk of the grid is dispatched.
    // The ranges of ONE sequence would be a whole group of workgroups apart -- with a group of 8 k workgroups every
    // range of a long sequence on the same XCD -- so each group is rotated by `group_rot` items against the one
    // before it (the host picks 0 or one workgroup's worth, whichever makes the distance odd: the ranges then
    // cycle through all XCDs).  No padding: a grid that fills the chip exactly keeps doing so.
    const int grp = item / p.group_items;
    int rest = item - grp * p.group_items;
    const int split = p.nsplit - 1 - grp;
    const int pairs = p.B * p.H_kv;
    if (split < 0 || rest >= pairs) return;          // behind the grid's end / the last workgroup's spare waves
    rest += (int)((unsigned)(grp * p.group_rot) % (unsigned)pairs);   // (grp < 2^31 / 4 items: no overflow)
    rest = rest >= pairs ? rest - pairs : rest;
    const int h = rest % p.H_kv;
    const int b = rest / p.H_kv;

    const int n_k = p.cu_k[b + 1] - p.cu_k[b];
    // FUSE: the token's position is read with the sequence's bounds (one scalar round trip for all three)
    const int64_t f_pos_raw = FUSE ? p.positions[b] : 0;
    int tq = b;
    if (p.cu_q) {
        if (p.cu_q[b + 1] - p.cu_q[b] != 1) return;  // not a decode row
        tq = p.cu_q[b];
    }
    const int kc0 = split * p.chunk;            // chunk of the work item
    if (kc0 >= n_k && !(split == 0)) return;    // uniform over the workgroup when WGS
    // this wave's keys [k0, k1): the whole chunk, or its wave-th quarter (a multiple of 16 keys)
    const int k0 = WGS ? kc0 + wave * (p.chunk / kDecWaves) : kc0;
    const int k1 = WGS ? min(min(kc0 + p.chunk, n_k), k0 + p.chunk / kDecWaves) : min(kc0 + p.chunk, n_k);
    const bool direct = p.nsplit == 1 || n_k <= p.chunk;   // a single key range: the output row is written at once

    uint4 qv[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
        qv[g] = p.q[(int64_t)tq * p.q_stride_vec + (int64_t)(h * G + g) * 16 + c];
    // FUSE: does this wave's key range hold the step's new token (position n_k - 1)?  Wave-uniform.
    // (A caller whose max_seqlen_k is smaller than this sequence's key count gets the keys behind the planned
    // range ignored by the attention, as in the separate launch; the cache must stay right all the same, so the
    // last work item of the range takes the token then.)
    const bool last_item = split == p.nsplit - 1 && (!WGS || wave == kDecWaves - 1);
    const bool own_new = FUSE && n_k - 1 >= k0 && (n_k - 1 < k1 || last_item);
    uint4 k_rot = make_uint4(0, 0, 0, 0), v_row = make_uint4(0, 0, 0, 0);
    uint4 f_qwa = k_rot, f_qwb = k_rot, f_kwa = k_rot, f_kwb = k_rot;
    float4 f_c0 = make_float4(0.f, 0.f, 0.f, 0.f), f_c1 = f_c0, f_s0 = f_c0, f_s1 = f_c0;
    if (FUSE) {
        // everything the rotation needs is requested here, in front of the first key tile; the rotation
        // itself runs behind that tile's requests
        if (own_new) {
            k_rot = p.k_new[(int64_t)tq * p.k_stride_vec + (int64_t)h * 16 + c];
            v_row = p.v_new[(int64_t)tq * p.v_stride_vec + (int64_t)h * 16 + c];
        }
        if (p.q_w != nullptr) {
            f_qwa = p.q_w[c & 7];
            f_qwb = p.q_w[8 + (c & 7)];
            f_kwa = p.k_w[c & 7];
            f_kwb = p.k_w[8 + (c & 7)];
        }
        // the position is a scalar read like the sequence's bounds above, so the cos / sin row can be
        // requested here as well: every request of the rotation then sits in front of the first key tile's,
        // and the rotation waits for them only (the memory counter retires in order)
        int64_t pos = f_pos_raw;           // (token b is sequence b's: this form takes no cu_seqlens_q)
        pos = pos < 0 ? 0 : (pos >= p.max_pos ? p.max_pos - 1 : pos);
        const float4* cs = p.cos_sin + pos * 32;
        f_c0 = cs[2 * (c & 7)];
        f_c1 = cs[2 * (c & 7) + 1];
        f_s0 = cs[16 + 2 * (c & 7)];
        f_s1 = cs[16 + 2 * (c & 7) + 1];
    }

    float m[G], l[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = kNegBig;
        l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
    }

    const int n_keys = k1 - k0;
    const int nt = n_keys > 0 ? (n_keys + 15) >> 4 : 0;
    const int32_t* bt = p.block_tables + (int64_t)b * p.max_blocks;
    const int row_vec = p.H_kv * 16;  // uint4 per cache token row

    // Load cursor: cache block index and offset of the next tile's first key.  With
    // block_size a multiple of 16 a 16-key tile never straddles cache blocks, so the
    // physical block id is one scalar table read per tile; other block sizes resolve the
    // block per lane.
    uint4 kbuf[kDepth + 1][4], vbuf[kDepth + 1][4];
    int cur_blk = k0 / p.block_size;
    int cur_off = k0 - cur_blk * p.block_size;
    int cur_key = k0;
    auto tile_load = [&](uint4 (&kd)[4], uint4 (&vd)[4]) __attribute__((always_inline)) {
        if (ALIGNED) {
            int64_t phys = __builtin_amdgcn_readfirstlane(bt[cur_blk]);
            phys = phys < 0 ? 0 : (phys >= p.num_blocks ? p.num_blocks - 1 : phys);
            const int64_t base = ((phys * p.block_size + cur_off) * p.H_kv + h) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = 4 * i + kq;
                const int jj = (cur_key + j) < k1 ? j : 0;  // tail rows re-read row 0; masked below
                const int64_t vec = base + jj * row_vec + c;
                kd[i] = kv_load(p.k_cache + vec);
                vd[i] = kv_load(p.v_cache + vec);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int key = cur_key + 4 * i + kq;
                key = key < k1 ? key : k1 - 1;
                const int blk = key / p.block_size;
                const int off = key - blk * p.block_size;
                int64_t phys = bt[blk];
                phys = phys < 0 ? 0 : (phys >= p.num_blocks ? p.num_blocks - 1 : phys);
                const int64_t vec = ((phys * p.block_size + off) * p.H_kv + h) * 16 + c;
                kd[i] = kv_load(p.k_cache + vec);
                vd[i] = kv_load(p.v_cache + vec);
            }
        }
        cur_key += 16;
        cur_off += 16;
        if (cur_off >= p.block_size) {
            cur_off -= p.block_size;
            ++cur_blk;
        }
    };

    // One tile: scores for 16 keys (4 per 16-lane row), online-softmax update, PV accumulate.
    auto tile_compute = [&](int t, uint4 (&kd)[4], uint4 (&vd)[4], auto may_patch) __attribute__((always_inline)) {
        if constexpr (FUSE && decltype(may_patch)::value) {
            // the tile that holds position n_k - 1 was fetched from a slot nobody has written yet (it may hold
            // anything, NaN bit patterns included).  The lanes behind the end of the keys re-read the tile's
            // first row, which is that slot when n_k - 1 is the tile's first key: they take the new rows too
            // (their probabilities are zero, but 0 x NaN is not)
            const int jn = n_k - 1 - (k0 + 16 * t);
            if (own_new && jn >= 0 && jn < 16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool sel = ((jn >> 2) == i && (jn & 3) == kq) || (k0 + 16 * t + 4 * i + kq) >= k1;
                    kd[i] = pick(sel, k_rot, kd[i]);
                    vd[i] = pick(sel, v_row, vd[i]);
                }
            }
        }
        float s[4][G];
        bool valid[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            valid[i] = (k0 + 16 * t + 4 * i + kq) < k1;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float d = row16_sum(dot8(kd[i], qv[g])) * p.scale_log2;
                s[i][g] = valid[i] ? d : kNegBig;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float mn = fmaxf(fmaxf(fmaxf(s[0][g], s[1][g]), fmaxf(s[2][g], s[3][g])), m[g]);
            const float alpha = __builtin_amdgcn_exp2f(m[g] - mn);
            m[g] = mn;
            float ps = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[i][g] = valid[i] ? __builtin_amdgcn_exp2f(s[i][g] - mn) : 0.f;
                ps += s[i][g];
            }
            l[g] = l[g] * alpha + ps;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[g][j] *= alpha;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t vu[4] = {vd[i].x, vd[i].y, vd[i].z, vd[i].w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float lo = bf16_lo(vu[w]), hi = bf16_hi(vu[w]);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    acc[g][2 * w] += s[i][g] * lo;
                    acc[g][2 * w + 1] += s[i][g] * hi;
                }
            }
        }
    };

    // Software pipeline over kDepth+1 register sets (statically indexed ring): the loads of
    // the next kDepth tiles are in flight while a tile is consumed.
#pragma unroll
    for (int d = 0; d < kDepth; ++d)
        if (d < nt) tile_load(kbuf[d], vbuf[d]);
    // FUSE without register prefetch: the first TWO tiles are requested here all the same (the second into
    // registers of its own that are dead afterwards), so that the rotation below runs under their round trip
    // and the wave has made up for it by the time it enters the loop
    uint4 kx[4], vx[4];
    const bool ahead1 = FUSE && kDepth == 0 && nt > 0, ahead2 = ahead1 && nt > 1;
    if (ahead1) tile_load(kbuf[0], vbuf[0]);
    if (ahead2) tile_load(kx, vx);
    if (FUSE) {
        // lane c holds piece c of a head: pieces 0..7 are x1 (first half), 8..15 are x2; head_norm_rope wants
        // (x1 piece c', x2 piece c') in each lane of an aligned group of 8, so a lane takes its partner's piece
        // from lane c ^ 8 (row_ror:8) and both halves of the 16-lane row compute the head; lanes c < 8 keep
        // the rotated x1 piece, the others the x2 piece
        const bool upper = c >= 8;
        const bool norm = p.q_w != nullptr;
        const float4 c0 = f_c0, c1 = f_c1, s0 = f_s0, s1 = f_s1;
        auto partner = [](const uint4& x) __attribute__((always_inline)) {
            uint4 r;
            r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.x, 0x128, 0xF, 0xF, true);   // row_ror:8
            r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.y, 0x128, 0xF, 0xF, true);
            r.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.z, 0x128, 0xF, 0xF, true);
            r.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.w, 0x128, 0xF, 0xF, true);
            return r;
        };
        auto rotate = [&](const uint4& own, const uint4& wa, const uint4& wb) __attribute__((always_inline)) {
            const uint4 oth = partner(own);
            uint4 o1, o2;
            const uint4 x1 = pick(upper, oth, own), x2 = pick(upper, own, oth);
            if (norm) head_norm_rope<true>(x1, x2, wa, wb, c0, c1, s0, s1, p.eps, o1, o2);
            else head_norm_rope<false>(x1, x2, wa, wb, c0, c1, s0, s1, p.eps, o1, o2);
            return pick(upper, o2, o1);
        };
        // Every 16-lane row of the wave holds the same G query heads (and the new key): row r rotates item r
        // of (query heads 0 .. G-1, key) and the results are passed round, one pass of the arithmetic instead
        // of G + 1 (the rotation's VALU time is in front of the wave's first tile: 2.1 us of the launch with
        // three passes at G = 2)
        static_assert(!FUSE || G + 1 <= 4, "one item per 16-lane row");
        uint4 item = k_rot, wa = f_kwa, wb = f_kwb;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            item = pick(kq == g, qv[g], item);
            wa = pick(kq == g, f_qwa, wa);
            wb = pick(kq == g, f_qwb, wb);
        }
        const uint4 done = rotate(item, wa, wb);
        auto from_row = [&](int r) __attribute__((always_inline)) {
            const int src = (16 * r + c) * 4;
            uint4 o;
            o.x = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)done.x);
            o.y = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)done.y);
            o.z = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)done.z);
            o.w = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)done.w);
            return o;
        };
#pragma unroll
        for (int g = 0; g < G; ++g) qv[g] = from_row(g);
        if (own_new) k_rot = from_row(G);
    }
    int t_first = 0;
    constexpr std::true_type kPatch{};
    constexpr std::false_type kNoPatch{};
    if (ahead1) {
        tile_compute(0, kbuf[0], vbuf[0], kPatch);
        t_first = 1;
    }
    if (ahead2) {
        tile_compute(1, kx, vx, kPatch);
        t_first = 2;
    }
    if constexpr (FUSE) {
        // position n_k - 1 can only lie in the wave's LAST tile: the groups of tiles in front of it run the
        // plain kernel's loop (no patch code, none of its registers live across the loop); measured 107.3 ->
        // 106.5 us on the 256-sequence ragged batch next to 106.0 for the attention launch alone
        int t = t_first;
        for (; t + kDepth + 1 <= nt - 1; t += kDepth + 1) {
#pragma unroll
            for (int u = 0; u <= kDepth; ++u) {
                if (t + u + kDepth < nt)
                    tile_load(kbuf[(u + kDepth) % (kDepth + 1)], vbuf[(u + kDepth) % (kDepth + 1)]);
                tile_compute(t + u, kbuf[u], vbuf[u], kNoPatch);
            }
        }
#pragma unroll
        for (int u = 0; u <= kDepth; ++u) {
            if (t + u < nt) {
                if (t + u + kDepth < nt)
                    tile_load(kbuf[(u + kDepth) % (kDepth + 1)], vbuf[(u + kDepth) % (kDepth + 1)]);
                tile_compute(t + u, kbuf[u], vbuf[u], kPatch);
            }
        }
    } else {
        for (int t = t_first; t < nt; t += kDepth + 1) {
#pragma unroll
            for (int u = 0; u <= kDepth; ++u) {
                if (t + u < nt) {
                    if (t + u + kDepth < nt)
                        tile_load(kbuf[(u + kDepth) % (kDepth + 1)], vbuf[(u + kDepth) % (kDepth + 1)]);
                    tile_compute(t + u, kbuf[u], vbuf[u], kPatch);
                }
            }
        }
    }
    if (FUSE && own_new && kq == 0) {
        // the new token's rows go to the cache behind the key stream (a store in front of it would hold back
        // every cache load behind it: the compiler has to assume they may alias)
        const int64_t slot = p.slots[tq];
        if (slot >= 0 && slot < p.num_slots) {
            p.k_cache_w[(slot * p.H_kv + h) * 16 + c] = k_rot;
            p.v_cache_w[(slot * p.H_kv + h) * 16 + c] = v_row;
        }
    }

    // Merge the four 16-lane rows of the wave (lanes l, l^16, l^32, l^48 hold the same dims).
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float ma = fmaxf(m[g], __shfl_xor(m[g], 16));
        ma = fmaxf(ma, __shfl_xor(ma, 32));
        const float f = __builtin_amdgcn_exp2f(m[g] - ma);
        float la = l[g] * f;
        la += __shfl_xor(la, 16);
        la += __shfl_xor(la, 32);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = acc[g][j] * f;
            a += __shfl_xor(a, 16);
            a += __shfl_xor(a, 32);
            acc[g][j] = a;
        }
        m[g] = ma;
        l[g] = la;
    }

    if (WGS) {
        wg_merge_store<G, TK>(p, wg_sm, wave, lane, b, h, split, tq, m, l, acc, direct);
        if (TK && !direct) wg_ticket_merge(p, reinterpret_cast<int*>(wg_sm), b, h, tq, n_k, wave, lane);
        return;
    }
    if (kq == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) store_state<TK>(p, b, h * G + g, split, tq, c, m[g], l[g], acc[g], direct);
    }
    if (TK && !direct) wave_ticket_merge(p, b, h, tq, n_k, lane);
}

// Variant for G >= 4 query heads per kv head.  The per-(key, head) lane reductions of the
// kernel above cost 5 VALU ops per score and make it instruction-bound once G reaches 4;
// here the scores of a 16-key tile come from four v_mfma_f32_16x16x32_bf16
// (S^T[key][q] = K . Q^T, K fragments loaded from global memory directly in operand shape,
// 64 contiguous bytes per key row and instruction, Q fragments resident, columns >= G zero).
// Lane (c = lane & 15, r = lane >> 4) then holds S^T[4r + i][q = c], i = 0..3: every score
// exists once per wave, so the exponentials are not replicated.  The probabilities go
// through a 512-byte per-wave LDS exchange into the V layout of the kernel above (lane =
// (key 4i + r of load i, 16-byte piece c)), where PV stays packed fp32 FMAs.
// The running maximum is uniform per query head across the wave and only raised when a
// score exceeds it by more than 2^kDecRaise (then the accumulators are rescaled), so the
// common tile has no cross-lane maximum at all.
constexpr float kDecRaise = 8.0f;

// G is the head count the kernel is laid out for (4 or 8); the launch may have fewer query heads per
// kv head (p.G = 3, 5, 6, 7: e.g. 40 query / 8 kv heads): the surplus columns are zero queries whose
// results are not stored.
//
// FUSE: as in attn_decode_kernel (raw projections in, the step's token normalised, rotated and stored by the
// launch).  The rotation runs in that kernel's layout -- lane (row, c) holds piece c of item 4 j + row of (query
// heads 0 .. p.G - 1, new key) in pass j -- and the rotated heads cross a per-wave LDS area into the operand
// layouts of this kernel: Q fragments for every lane, the new key's fragments for the lanes of its tile row.
template <int G, bool ALIGNED, int kDepth, bool WGS, bool TK, bool FUSE = false>
__global__ __launch_bounds__(kDecWaves * 64) void attn_decode_mfma_kernel(const DecodeParams p) {
    static_assert(G == 4 || G == 8, "query heads per kv head");
    constexpr int kFusePasses = (G + 1 + 3) / 4;      // items (query heads + key) over the wave's four rows
    __shared__ uint4 fstage_all[FUSE ? kDecWaves : 1][FUSE ? (G + 1) * 16 : 1];
    __shared__ __attribute__((aligned(16))) float xch_all[kDecWaves][16 * G + 16];
    // workgroup items: the waves' states (its first word later carries the arrival number of the in-launch combine)
    __shared__ __attribute__((aligned(16))) float wg_sm[WGS ? kDecWaves * G * kPartStride : 4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r4 = lane >> 4;  // S^T: key group; V layout: key of a 4-key load
    const int c = lane & 15;   // S^T: query head; V layout: 16-byte piece of the head row
    float* xp = xch_all[wave];       // [16 keys][G] probabilities of the current tile
    float* xa = xch_all[wave] + 16 * G;  // [16] per-head scalars (alpha, then m, l)

    const int item = WGS ? (int)blockIdx.x : blockIdx.x * kDecWaves + wave;
    // Key-range-major: the items of the batch's LAST key range first, ..., those of range 0 last; within a range
    // (sequence, kv head) with the head fastest.  A batch of short sequences with one long one has work in range 0
    // only, except for that sequence: sequence-major order would put the few non-empty workgroups at a fixed stride
    // in the grid -- always the same XCDs and CUs under round-robin dispatch -- with empty ones between; last
    // ranges first puts the longest sequences' extra work in flight before the bulk of the grid is dispatched.
    // The ranges of ONE sequence would be a whole group of workgroups apart -- with a group of 8 k workgroups every
    // range of a long sequence on the same XCD -- so each group is rotated by `group_rot` items against the one
    // before it (the host picks 0 or one workgroup's worth, whichever makes the distance odd: the ranges then
    // cycle through all XCDs).  No padding: a grid that fills the chip exactly keeps doing so.
    const int grp = item / p.group_items;
    int rest = item - grp * p.group_items;
    const int split = p.nsplit - 1 - grp;
    const int pairs = p.B * p.H_kv;
    if (split < 0 || rest >= pairs) return;          // behind the grid's end / the last workgroup's spare waves
    rest += (int)((unsigned)(grp * p.group_rot) % (unsigned)pairs);   // (grp < 2^31 / 4 items: no overflow)
    rest = rest >= pairs ? rest - pairs : rest;
    const int h = rest % p.H_kv;
    const int b = rest / p.H_kv;

    const int n_k = p.cu_k[b + 1] - p.cu_k[b];
    int tq = b;
    if (p.cu_q) {
        if (p.cu_q[b + 1] - p.cu_q[b] != 1) return;  // not a decode row
        tq = p.cu_q[b];
    }
    const int kc0 = split * p.chunk;            // chunk of the work item
    if (kc0 >= n_k && !(split == 0)) return;    // uniform over the workgroup when WGS
    // this wave's keys [k0, k1): the whole chunk, or its wave-th quarter (a multiple of 16 keys)
    const int k0 = WGS ? kc0 + wave * (p.chunk / kDecWaves) : kc0;
    const int k1 = WGS ? min(min(kc0 + p.chunk, n_k), k0 + p.chunk / kDecWaves) : min(kc0 + p.chunk, n_k);
    const bool direct = p.nsplit == 1 || n_k <= p.chunk;   // a single key range: the output row is written at once

    // B operand: lane (c, r4) holds Q[head c][dims 32 s + 8 r4 .. +7] for k-step s
    bf16x8 qf[4];
    if (!FUSE) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < p.G) v = p.q[(int64_t)tq * p.q_stride_vec + (int64_t)(h * p.G + c) * 16 + 4 * s + r4];
            qf[s] = __builtin_bit_cast(bf16x8, v);
        }
    }
    // FUSE: the raw rows of the wave's items, the value row, the norm weights and the cos / sin row are all
    // requested here, in front of the first key tile (see attn_decode_kernel)
    const bool last_item = split == p.nsplit - 1 && (!WGS || wave == kDecWaves - 1);
    const bool own_new = FUSE && n_k - 1 >= k0 && (n_k - 1 < k1 || last_item);
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    uint4 f_raw[kFusePasses], v_row = zero4, f_qwa = zero4, f_qwb = zero4, f_kwa = zero4, f_kwb = zero4;
    float4 f_c0 = make_float4(0.f, 0.f, 0.f, 0.f), f_c1 = f_c0, f_s0 = f_c0, f_s1 = f_c0;
    if (FUSE) {
#pragma unroll
        for (int j = 0; j < kFusePasses; ++j) {
            const int it = 4 * j + r4;
            const uint4* src = it < p.G ? p.q + (int64_t)tq * p.q_stride_vec + (int64_t)(h * p.G + it) * 16 + c
                                        : p.k_new + (int64_t)tq * p.k_stride_vec + (int64_t)h * 16 + c;
            uint4 x = zero4;
            if (it < p.G || (it == p.G && own_new)) x = *src;
            f_raw[j] = x;
        }
        if (own_new) v_row = p.v_new[(int64_t)tq * p.v_stride_vec + (int64_t)h * 16 + c];
        if (p.q_w != nullptr) {
            f_qwa = p.q_w[c & 7];
            f_qwb = p.q_w[8 + (c & 7)];
            f_kwa = p.k_w[c & 7];
            f_kwb = p.k_w[8 + (c & 7)];
        }
        int64_t pos = p.positions[b];      // (token b is sequence b's: this form takes no cu_seqlens_q)
        pos = pos < 0 ? 0 : (pos >= p.max_pos ? p.max_pos - 1 : pos);
        const float4* cs = p.cos_sin + pos * 32;
        f_c0 = cs[2 * (c & 7)];
        f_c1 = cs[2 * (c & 7) + 1];
        f_s0 = cs[16 + 2 * (c & 7)];
        f_s1 = cs[16 + 2 * (c & 7) + 1];
    }
    uint4 k_frag_new[4] = {zero4, zero4, zero4, zero4}, k_store = zero4;   // FUSE: the new key in both layouts

    float m_run = kNegBig, l_part = 0.f;
    float acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;

    const int n_keys = k1 - k0;
    const int nt = n_keys > 0 ? (n_keys + 15) >> 4 : 0;
    const int32_t* bt = p.block_tables + (int64_t)b * p.max_blocks;
    const int row_vec = p.H_kv * 16;

    uint4 kbuf[kDepth + 1][4], vbuf[kDepth + 1][4];
    int cur_blk = k0 / p.block_size;
    int cur_off = k0 - cur_blk * p.block_size;
    int cur_key = k0;
    auto resolve = [&](int key) -> int64_t {  // first uint4 of this head's row of a logical key
        key = key < k1 ? key : k1 - 1;
        const int blk = key / p.block_size;
        const int off = key - blk * p.block_size;
        int64_t phys = bt[blk];
        phys = phys < 0 ? 0 : (phys >= p.num_blocks ? p.num_blocks - 1 : phys);
        return ((phys * p.block_size + off) * p.H_kv + h) * 16;
    };
    auto tile_load = [&](uint4 (&kd)[4], uint4 (&vd)[4]) {
        if (ALIGNED) {
            int64_t phys = __builtin_amdgcn_readfirstlane(bt[cur_blk]);
            phys = phys < 0 ? 0 : (phys >= p.num_blocks ? p.num_blocks - 1 : phys);
            const int64_t base = ((phys * p.block_size + cur_off) * p.H_kv + h) * 16;
            const int kr = (cur_key + c) < k1 ? c : 0;  // A operand row = key c of the tile
            const int64_t kvec = base + kr * row_vec + r4;
#pragma unroll
            for (int s = 0; s < 4; ++s) kd[s] = kv_load(p.k_cache + kvec + 4 * s);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = 4 * i + r4;
                const int jj = (cur_key + j) < k1 ? j : 0;
                vd[i] = kv_load(p.v_cache + base + jj * row_vec + c);
            }
        } else {
            const int64_t kvec = resolve(cur_key + c) + r4;
#pragma unroll
            for (int s = 0; s < 4; ++s) kd[s] = kv_load(p.k_cache + kvec + 4 * s);
#pragma unroll
            for (int i = 0; i < 4; ++i) vd[i] = kv_load(p.v_cache + resolve(cur_key + 4 * i + r4) + c);
        }
        cur_key += 16;
        cur_off += 16;
        if (cur_off >= p.block_size) {
            cur_off -= p.block_size;
            ++cur_blk;
        }
    };

    // G consecutive floats of the exchange area (the widest LDS reads that fit)
    auto read_heads = [&](const float* src, float (&dst)[G]) {
        if constexpr (G >= 4) {
#pragma unroll
            for (int g4 = 0; g4 < G / 4; ++g4) {
                const float4 a = *reinterpret_cast<const float4*>(src + 4 * g4);
                dst[4 * g4] = a.x;
                dst[4 * g4 + 1] = a.y;
                dst[4 * g4 + 2] = a.z;
                dst[4 * g4 + 3] = a.w;
            }
        } else if constexpr (G == 2) {
            const float2 a = *reinterpret_cast<const float2*>(src);
            dst[0] = a.x;
            dst[1] = a.y;
        } else {
            dst[0] = src[0];
        }
    };

    auto tile_compute = [&](int t, uint4 (&kd)[4], uint4 (&vd)[4], auto may_patch) __attribute__((always_inline)) {
        if constexpr (FUSE && decltype(may_patch)::value) {
            // the tile that holds position n_k - 1 was fetched from a slot nobody has written yet: the lanes of
            // its key row take the new key's fragments, the lanes of its value row -- and those behind the end
            // of the keys, which re-read the tile's first row -- the new value row (0 x NaN is not 0; a stale K
            // row only reaches scores that are masked by a select)
            const int jn = n_k - 1 - (k0 + 16 * t);
            if (own_new && jn >= 0 && jn < 16) {
#pragma unroll
                for (int s = 0; s < 4; ++s) kd[s] = pick(c == jn, k_frag_new[s], kd[s]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool sel = (4 * i + r4) == jn || (k0 + 16 * t + 4 * i + r4) >= k1;
                    vd[i] = pick(sel, v_row, vd[i]);
                }
            }
        }
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kd[s]), qf[s], sacc, 0, 0, 0);
        float sc[4];
        const int key_base = k0 + 16 * t + 4 * r4;
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[i] = (key_base + i) < k1 ? sacc[i] * p.scale_log2 : kNegBig;
        const float lmax = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        if (__builtin_amdgcn_ballot_w64(lmax > m_run + kDecRaise) != 0) {
            // raise the running maxima (uniform over the four key groups) and rescale
            float tm = fmaxf(lmax, __shfl_xor(lmax, 16));
            tm = fmaxf(tm, __shfl_xor(tm, 32));
            const float m_new = fmaxf(m_run, tm);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_part *= alpha;
            if (r4 == 0) xa[c] = alpha;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float av[G];
            read_heads(xa, av);
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[g][j] *= av[g];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sc[i] = __builtin_amdgcn_exp2f(sc[i] - m_run);  // masked keys: exp2(-1e30 - m) = 0
            ps += sc[i];
        }
        l_part += ps;
        if (c < G) {
#pragma unroll
            for (int i = 0; i < 4; ++i) xp[(4 * r4 + i) * G + c] = sc[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float pg[G];
            read_heads(xp + (4 * i + r4) * G, pg);
            const uint32_t vu[4] = {vd[i].x, vd[i].y, vd[i].z, vd[i].w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float lo = bf16_lo(vu[w]), hi = bf16_hi(vu[w]);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    acc[g][2 * w] += pg[g] * lo;
                    acc[g][2 * w + 1] += pg[g] * hi;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

#pragma unroll
    for (int d = 0; d < kDepth; ++d)
        if (d < nt) tile_load(kbuf[d], vbuf[d]);
    if (FUSE) {
        uint4* stage = fstage_all[wave];
        const bool upper = c >= 8;
        const bool norm = p.q_w != nullptr;
#pragma unroll
        for (int j = 0; j < kFusePasses; ++j) {
            const int it = 4 * j + r4;
            const bool is_q = it < p.G;
            const uint4 own = f_raw[j];
            uint4 oth;
            oth.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own.x, 0x128, 0xF, 0xF, true);   // row_ror:8
            oth.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own.y, 0x128, 0xF, 0xF, true);
            oth.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own.z, 0x128, 0xF, 0xF, true);
            oth.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own.w, 0x128, 0xF, 0xF, true);
            const uint4 x1 = pick(upper, oth, own), x2 = pick(upper, own, oth);
            const uint4 wa = pick(is_q, f_qwa, f_kwa), wb = pick(is_q, f_qwb, f_kwb);
            uint4 o1, o2;
            if (norm) head_norm_rope<true>(x1, x2, wa, wb, f_c0, f_c1, f_s0, f_s1, p.eps, o1, o2);
            else head_norm_rope<false>(x1, x2, wa, wb, f_c0, f_c1, f_s0, f_s1, p.eps, o1, o2);
            if (it <= p.G) stage[it * 16 + c] = pick(upper, o2, o1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 v = zero4;
            if (c < p.G) v = stage[c * 16 + 4 * s + r4];
            qf[s] = __builtin_bit_cast(bf16x8, v);
            k_frag_new[s] = stage[p.G * 16 + 4 * s + r4];
        }
        k_store = stage[p.G * 16 + c];
    }
    // (peeling the patched last tile off the loop, as attn_decode_kernel does, measured 1 - 4 % slower here)
    constexpr std::true_type kPatch{};
    for (int t = 0; t < nt; t += kDepth + 1) {
#pragma unroll
        for (int u = 0; u <= kDepth; ++u) {
            if (t + u < nt) {
                if (t + u + kDepth < nt) tile_load(kbuf[(u + kDepth) % (kDepth + 1)], vbuf[(u + kDepth) % (kDepth + 1)]);
                tile_compute(t + u, kbuf[u], vbuf[u], kPatch);
            }
        }
    }
    if (FUSE && own_new && r4 == 0) {
        // the new token's rows go to the cache behind the key stream (see attn_decode_kernel)
        const int64_t slot = p.slots[b];
        if (slot >= 0 && slot < p.num_slots) {
            p.k_cache_w[(slot * p.H_kv + h) * 16 + c] = k_store;
            p.v_cache_w[(slot * p.H_kv + h) * 16 + c] = v_row;
        }
    }

    // Denominators: sum the four key groups, then hand (m, l) per head to the V-layout lanes.
    float l_tot = l_part + __shfl_xor(l_part, 16);
    l_tot += __shfl_xor(l_tot, 32);
    if (r4 == 0) xa[c] = l_tot;
    if (r4 == 1) xp[c] = m_run;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float l[G], m[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        l[g] = xa[g];
        m[g] = xp[g];
    }
    // All four key rows share one reference maximum: plain sums merge them.
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = acc[g][j];
            a += __shfl_xor(a, 16);
            a += __shfl_xor(a, 32);
            acc[g][j] = a;
        }

    if (WGS) {
        wg_merge_store<G, TK>(p, wg_sm, wave, lane, b, h, split, tq, m, l, acc, direct);
        if (TK && !direct) wg_ticket_merge(p, reinterpret_cast<int*>(wg_sm), b, h, tq, n_k, wave, lane);
        return;
    }
    if (r4 == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (g < p.G) store_state<TK>(p, b, h * p.G + g, split, tq, c, m[g], l[g], acc[g], direct);
    }
    if (TK && !direct) wave_ticket_merge(p, b, h, tq, n_k, lane);
}

// Combine the per-chunk partial states of one (sequence, query head): 32 threads x 4 dims.
// SG = 1: a 256-thread block handles 8 pairs, each 32-thread group walks all chunks of its pair
// (few chunks).  SG = 8: a block handles one pair and its 8 groups walk every 8th chunk, then
// reduce through LDS (many chunks: the walk is a chain of dependent-latency loads).
template <int SG>
__global__ __launch_bounds__(256) void attn_decode_merge_kernel(const DecodeParams p, int H_q) {
    __shared__ float red_m[8];
    __shared__ float red_den[8];
    __shared__ float4 red_num[8][32];
    const int grp = threadIdx.x >> 5;
    const int d4 = threadIdx.x & 31;
    const int pair = SG == 1 ? blockIdx.x * 8 + grp : blockIdx.x;
    if (pair >= p.B * H_q) return;  // SG == 8: uniform per block; SG == 1: no barriers below
    const int b = pair / H_q;
    const int hq = pair - b * H_q;
    const int n_k = p.cu_k[b + 1] - p.cu_k[b];
    int tq = b;
    if (p.cu_q) {
        if (p.cu_q[b + 1] - p.cu_q[b] != 1) return;  // uniform per block in both forms
        tq = p.cu_q[b];
    }
    if (n_k <= p.chunk) return;   // a single key range: written by the attention launch itself (uniform per block)
    int ns = (n_k + p.chunk - 1) / p.chunk;
    ns = ns > p.nsplit ? p.nsplit : ns;   // (a caller's max_seqlen_k below this sequence's key count)
    const float* base = p.part + ((int64_t)b * H_q + hq) * p.nsplit * kPartStride;
    const int s0 = SG == 1 ? 0 : grp;
    float mmax = kNegBig;
#pragma unroll 4
    for (int s = s0; s < ns; s += SG) mmax = fmaxf(mmax, base[s * kPartStride + kHeadDim]);
    if (SG > 1) {
        if (d4 == 0) red_m[grp] = mmax;
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 8; ++g) mmax = fmaxf(mmax, red_m[g]);
    }
    float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
    float den = 0.f;
#pragma unroll 4
    for (int s = s0; s < ns; s += SG) {
        const float2 ml = *reinterpret_cast<const float2*>(base + s * kPartStride + kHeadDim);
        const float4 a = *reinterpret_cast<const float4*>(base + s * kPartStride + 4 * d4);
        const float f = __builtin_amdgcn_exp2f(ml.x - mmax);
        num.x += a.x * f;
        num.y += a.y * f;
        num.z += a.z * f;
        num.w += a.w * f;
        den += ml.y * f;
    }
    if (SG > 1) {
        red_num[grp][d4] = num;
        if (d4 == 0) red_den[grp] = den;
        __syncthreads();
        if (grp != 0) return;
        num = make_float4(0.f, 0.f, 0.f, 0.f);
        den = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 a = red_num[g][d4];
            num.x += a.x;
            num.y += a.y;
            num.z += a.z;
            num.w += a.w;
            den += red_den[g];
        }
    }
    const float r = den > 0.f ? 1.0f / den : 0.f;
    uint2 w;
    w.x = pack_bf16(num.x * r, num.y * r);
    w.y = pack_bf16(num.z * r, num.w * r);
    __bf16* o = reinterpret_cast<__bf16*>(p.o) + (int64_t)tq * p.o_stride_vec * 8 + hq * kHeadDim;
    *reinterpret_cast<uint2*>(o + 4 * d4) = w;
}

// Launch plan of one call: which kernel, how the keys are cut into work items.  Computed from the
// shapes and the process-wide overrides (nv_set_tuning) only, so nv_attn_decode_workspace_bytes and
// the launch agree by construction.
//
// Keys per work item: smaller chunks give more, finer-grained waves (better balance on ragged
// batches, more loads in flight for small batches) at the price of more partials.
// G >= 4 runs the MFMA-score kernel (about 200 registers: one 4-wave workgroup per SIMD set and
// CU at a time), which measured fastest with one workgroup per CU; the VALU kernel wants two.
// Split launches lay the workspace out as [arrival counters: kTicketRegion bytes][partial states].
// The counter block has a fixed place and size so that memory the library owns (the built-in scratch)
// never holds anything else there: it starts at zero and every launch leaves it at zero.
constexpr int64_t kTicketRegion = 64 << 10;

struct DecodePlan {
    int chunk;        // keys per work item
    int nsplit;       // work items per (sequence, kv head)
    bool wgs;         // work item = workgroup (its waves split the chunk) instead of wave
    bool mfma;        // scores on the MFMA pipe (G >= 3)
    bool mfma_deep;   // MFMA-score kernel: one tile of register prefetch
    bool lean;        // VALU kernel: no register prefetch
};

static bool decode_uses_mfma(int G, const nv_tuning& tn) {
    if (G == 3 || (G > 4 && G < 8)) return true;  // only the MFMA kernel takes a padded head count
    if (tn.decode_score_path >= 0) return G >= 4 && tn.decode_score_path == 1;
    return G >= 4;
}

// chunk_keys > 0: the caller's choice of keys per work item (nv_attn_decode_paged_chunked: a host that knows the
// batch's lengths, not only the longest); the process-wide override still wins.
static DecodePlan plan_decode(int B, int H_kv, int max_k, int G, int chunk_keys = 0) {
    const nv_tuning tn = tuning();
    DecodePlan d;
    d.mfma = decode_uses_mfma(G, tn);
    const long long bh = (long long)B * H_kv;
    const long long target = tn.decode_target_waves > 0
                                 ? tn.decode_target_waves
                                 : (d.mfma ? NV_DEC_TARGET_WAVES / 2 : NV_DEC_TARGET_WAVES);
    // Few (sequence, kv head) pairs: a workgroup per work item, 4 waves on its chunk, so the
    // launch reaches the target with a quarter of the partial states (often none at all).
    // Not when the waves would stream >= 1024 keys each anyway and the cache row holds >= 4 kv
    // heads: there the partials are cheap, and four waves on neighbouring heads of the same
    // tokens (1 KiB contiguous per token) beat four waves on distant tokens (measured 6.2 vs
    // 5.7 TB/s at 8 sequences x 32 k keys).
    const long long wave_keys = max_k / ((target + bh - 1) / bh > 0 ? (target + bh - 1) / bh : 1);
    d.wgs = tn.decode_work_item >= 0 ? tn.decode_work_item == 1
                                     : (bh * 2 <= target && (H_kv <= 2 || wave_keys < 1024));
    if (tn.decode_chunk > 0) {
        d.chunk = tn.decode_chunk;
    } else if (chunk_keys > 0) {
        // a multiple of 64 keys (of 256 for workgroup items), and at most 256 ranges per sequence
        const long long unit = d.wgs ? 64 * kDecWaves : 64;
        long long chunk = ((long long)chunk_keys + unit - 1) / unit * unit;
        const long long least = (((long long)max_k + 255) / 256 + unit - 1) / unit * unit;
        d.chunk = (int)(chunk < least ? least : chunk);
    } else {
        // Fewest splits that still give the launch about `target` waves; unsplit sequences are
        // written directly (no partials, no merge launch).  Chunks are capped so that one long
        // sequence in a short batch cannot become a single-item tail.
        const long long want_items = d.wgs ? target / kDecWaves : target;
        long long want = (want_items + bh - 1) / bh;
        if (want < 1) want = 1;
        long long chunk = (max_k + want - 1) / want;
        chunk = (chunk + 63) / 64 * 64;
        // (2048 keys per wave at most: in a batch of short sequences with a few long ones the long ones' waves are
        // the launch's tail -- one 8192-key sequence among 256 of ~640 keys: 279 us at 4096 keys per wave, 237 at
        // 2048 -- while sequences of up to one chunk are written directly and never see the merge)
        long long lo = d.wgs ? 64 * kDecWaves : 64;
        const long long hi = d.wgs ? 2048 * kDecWaves : 2048;
        // one or a few short sequences: with 64 keys per wave the launch has fewer workgroups than the chip has
        // CUs, and its length is a wave's start-up plus its tiles one after the other; 32 keys per wave then
        // (one sequence of 4096 keys, 16 / 8 heads: 18.2 -> 15.5 us; 8 / 1 heads: 19.6 -> 15.1)
        if (d.wgs && bh * ((max_k + lo - 1) / lo) < 256) lo = 32 * kDecWaves;
        chunk = chunk < lo ? lo : (chunk > hi ? hi : chunk);
        d.chunk = (int)chunk;
    }
    const int n = (max_k + d.chunk - 1) / d.chunk;
    d.nsplit = n < 1 ? 1 : n;
    // Prefetch depth: one tile ahead hides HBM latency when each wave streams a long chunk
    // and waves are scarce; with many resident waves (or G = 8, where registers are the
    // limit) thread-level parallelism already covers it and the leaner kernel is faster.
    // (the lean kernel for split launches of many pairs as well measured 4 - 15 % slower on skewed batches)
    // A caller's chunk says that the lengths are uneven: then the waves of the long sequences run on after the
    // short ones' are gone, with less and less parallelism to hide latency behind, and the prefetch pays (rows
    // longest first, 16 / 8 heads, lengths spread over 100 .. 2000 keys, without / with prefetch: 256 sequences
    // 205.7 / 194.6 us, 512: 351.5 / 341.1, 1024: 659.7 / 649.6; equal lengths the other way round: 256 x 1024
    // keys 165.7 / 171.0, 256 x 2048: 325.0 / 328.8).
    d.lean = G >= 8 || (d.nsplit == 1 && bh >= NV_DEC_TARGET_WAVES && chunk_keys <= 0);
    // one tile of prefetch measured faster than none on every G >= 4 shape tried
    d.mfma_deep = tn.decode_mfma_depth >= 0 ? tn.decode_mfma_depth == 1 : true;
    return d;
}

}  // namespace nv

extern "C" int64_t nv_attn_decode_workspace_bytes_chunked(int B, int H_q, int H_kv, int D, int max_seqlen_k,
                                                          int chunk_keys) {
    using namespace nv;
    if (B <= 0 || H_q <= 0 || H_kv <= 0 || D != kHeadDim || max_seqlen_k < 0) return 0;
    const int ns = plan_decode(B, H_kv, max_seqlen_k, H_q / H_kv, chunk_keys).nsplit;
    if (ns == 1) return 16;
    return kTicketRegion + (int64_t)B * H_q * ns * kPartStride * (int64_t)sizeof(float);
}

extern "C" int64_t nv_attn_decode_workspace_bytes(int B, int H_q, int H_kv, int D, int max_seqlen_k) {
    return nv_attn_decode_workspace_bytes_chunked(B, H_q, H_kv, D, max_seqlen_k, 0);
}

// Keys per work item for a batch whose lengths the host knows (their sum beside their maximum); 0 = the plan's
// own choice stands.  The plan sees only the longest sequence: it cannot tell 256 sequences of 2048 keys (no
// split wanted: 4 - 9 % slower with one) from 255 of 640 and one of 8192 (whose waves are the launch's tail
// unless its keys are spread over many work items).  Measured on gfx950 (256 sequences of 513 .. 767 keys plus
// the long ones, 8 kv heads): one 8192-key sequence 225 us with the plan's 2048-key chunks, 153 with 512;
// four of 4096 keys 217 -> 145; one of 32768 keys 276 -> 249 (1024-key chunks: more ranges cost more empty
// work items than they save).
extern "C" int nv_attn_decode_suggest_chunk(int B, int H_q, int H_kv, int max_seqlen_k, int64_t sum_seqlen_k) {
    using namespace nv;
    if (B <= 0 || H_q <= 0 || H_kv <= 0 || H_q % H_kv != 0 || max_seqlen_k <= 0 || sum_seqlen_k <= 0) return 0;
    const DecodePlan d = plan_decode(B, H_kv, max_seqlen_k, H_q / H_kv);
    const long long bh = (long long)B * H_kv;
    const long long target = d.mfma ? NV_DEC_TARGET_WAVES / 2 : NV_DEC_TARGET_WAVES;
    const long long mean = (sum_seqlen_k + B - 1) / B;
    if (bh < target) {
        // (64 sequences, rows longest first, hinted against not: lengths spread over 100 .. 2000 keys, 1.9 x the
        // mean, 66 against 60 us; 513 .. 767 keys and one of 1500, 2.3 x: 44 / 43; one of 2000, 3.1 x: 44 / 51)
        if ((long long)max_seqlen_k * 2 < mean * 5) return 0;
        // Few pairs: the plan cuts every sequence into max / chunk ranges so that the launch has about `target`
        // waves -- which it only reaches when the sequences are equally long (the short ones of an uneven batch
        // have one range each).  With the sum known: as many keys per work item as make the launch full.
        const long long unit = d.wgs ? 64 * kDecWaves : 64;
        const long long items = d.wgs ? target / kDecWaves : target;
        long long chunk = (sum_seqlen_k * H_kv / items + unit - 1) / unit * unit;
        const long long least = d.wgs ? 512 : 256;
        if (chunk < least) chunk = least;
        return chunk * 4 <= (long long)d.chunk * 3 ? (int)chunk : 0;      // only when it is a real change
    }
    // Many more pairs than the target: a work item is one wave, every wave gets an equal share of the chip while
    // the others run and streams about 9 GB/s (18 keys per us) once it is alone, so a long sequence -- at most
    // 2048 keys per wave -- extends the launch by what it has left when the others are done.  Cutting everything
    // into pieces costs 3 - 8 % of the whole launch (partial states, merge launch, empty work items), which
    // grows with the launch while the extension does not: only when the longest sequence is at least half of
    // sum x H_kv / 256 keys.  Rows longest first, hinted against not: 512 sequences of 100 .. 2000 keys 378 / 349
    // us, with one of 6000 among them (0.36) 393 / 357, one of 10000 (0.59) 397 / 390; 512 of 513 .. 767 keys
    // and one of 6000 (0.58) 243 / 245, one of 8192 (0.78) 260 / 286; 32 / 8 heads, 256 of 100 .. 2000 keys
    // 225 / 208, 256 of 513 .. 767 and one of 3000 (0.57) 160 / 183.
    // What such a launch can still use is the prefetch that the plan leaves out for many equal sequences: when
    // the lengths are spread out (longest from 1.7 x the mean and 1024 keys on) the answer is the plan's own cap
    // of 2048 keys, which cuts nothing there and tells the plan that the lengths are uneven.
    if (bh * 4 > target * 5 && (long long)max_seqlen_k * 256 * 2 < sum_seqlen_k * H_kv)
        return (!d.mfma && d.nsplit == 1 && max_seqlen_k > 1024 && (long long)max_seqlen_k * 10 >= mean * 17) ? 2048 : 0;
    // A launch that is resident at once ends with its longest sequence (256 sequences of 513 .. 767 keys and
    // one of 1500: 155 us, 133 with 512-key pieces).  Balanced enough: the longest sequence within 1.4 x the
    // mean, or less than 512 keys beyond it (a split costs a merge launch and partial states: about 6 us)
    if ((long long)max_seqlen_k * 5 < mean * 7 || max_seqlen_k - mean < 512) return 0;
    // pieces of about three quarters of a mean sequence; twice that for the MFMA-score kernels, whose partial
    // states hold 4 - 8 query heads per kv head (64 / 8 heads, one 8192-key sequence among 256 of ~640 keys:
    // 172 us with 512-key pieces, 155 with 1024, 192 planned from the longest length alone)
    long long chunk = ((d.mfma ? mean * 3 / 2 : mean * 3 / 4) + 63) / 64 * 64;
    if (chunk < 256) chunk = 256;
    const long long most = 32;                                  // ranges per sequence (empty work items are not free)
    if (((long long)max_seqlen_k + chunk - 1) / chunk > most) chunk = ((max_seqlen_k + most - 1) / most + 63) / 64 * 64;
    return chunk < d.chunk ? (int)chunk : 0;
}

namespace nv {
// What the FUSE instantiations take beside the attention arguments (nv_attn_decode_paged_fused).
struct FusedDecodeIn {
    const void* k_new;
    const void* v_new;
    const void* q_w;
    const void* k_w;
    const float* cos_sin;
    const int64_t* positions;
    const int32_t* slots;
    int64_t k_stride_t, v_stride_t, max_pos;
    float eps;
};

static int decode_launch(const void* q, const void* k_cache, const void* v_cache, void* o,
                         const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                         const int32_t* block_tables, int B, int max_blocks, int max_seqlen_k, int H_q,
                         int H_kv, int D, int64_t num_blocks, int block_size, float scale,
                         int64_t q_stride_t, int64_t o_stride_t, void* workspace,
                         int64_t workspace_bytes, bool library_workspace, void* stream,
                         const FusedDecodeIn* fused = nullptr, int chunk_keys = 0, int flags = 0) {
    NV_REQUIRE(B >= 0 && max_blocks > 0 && max_seqlen_k >= 0 && H_q > 0 && H_kv > 0 &&
                   num_blocks > 0 && block_size > 0,
               NV_ERR_INVALID_ARG, "nv_attn_decode_paged: bad sizes B=%d max_blocks=%d max_k=%d", B,
               max_blocks, max_seqlen_k);
    NV_REQUIRE(D == kHeadDim, NV_ERR_UNSUPPORTED, "nv_attn_decode_paged: head_dim %d (128 only)", D);
    NV_REQUIRE(H_q % H_kv == 0, NV_ERR_INVALID_ARG, "nv_attn_decode_paged: H_q %% H_kv != 0");
    if (B == 0) return NV_OK;
    NV_REQUIRE(q && k_cache && v_cache && o && cu_seqlens_k && block_tables, NV_ERR_INVALID_ARG,
               "nv_attn_decode_paged: null pointer");
    NV_REQUIRE(q_stride_t % 8 == 0 && o_stride_t % 8 == 0 && q_stride_t >= (int64_t)H_q * D &&
                   o_stride_t >= (int64_t)H_q * D,
               NV_ERR_INVALID_ARG, "nv_attn_decode_paged: bad token strides");
    NV_REQUIRE(((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)o |
                (uintptr_t)workspace) % 16 == 0,
               NV_ERR_INVALID_ARG, "nv_attn_decode_paged: pointers must be 16-byte aligned");
    NV_REQUIRE((int64_t)max_blocks * block_size >= max_seqlen_k, NV_ERR_INVALID_ARG,
               "nv_attn_decode_paged: block table too narrow for max_seqlen_k");
    const int G = H_q / H_kv;
    NV_REQUIRE(G >= 1 && G <= 8, NV_ERR_UNSUPPORTED,
               "nv_attn_decode_paged: H_q/H_kv = %d (supported: 1 .. 8)", G);

    DecodeParams p;
    p.q = (const uint4*)q;
    p.k_cache = (const uint4*)k_cache;
    p.v_cache = (const uint4*)v_cache;
    p.o = (uint4*)o;
    p.cu_q = cu_seqlens_q;
    p.cu_k = cu_seqlens_k;
    p.block_tables = block_tables;
    p.part = nullptr;
    p.tickets = nullptr;
    p.B = B;
    p.max_blocks = max_blocks;
    p.H_kv = H_kv;
    p.block_size = block_size;
    NV_REQUIRE(chunk_keys >= 0, NV_ERR_INVALID_ARG, "nv_attn_decode_paged: chunk_keys %d < 0", chunk_keys);
    const DecodePlan plan = plan_decode(B, H_kv, max_seqlen_k, G, chunk_keys);
    p.chunk = plan.chunk;
    p.nsplit = plan.nsplit;
    p.G = G;
    p.num_blocks = num_blocks;
    p.q_stride_vec = q_stride_t / 8;
    p.o_stride_vec = o_stride_t / 8;
    p.scale_log2 = scale * 1.4426950408889634f;
    p.k_new = p.v_new = p.q_w = p.k_w = nullptr;
    p.cos_sin = nullptr;
    p.positions = nullptr;
    p.slots = nullptr;
    p.k_cache_w = p.v_cache_w = nullptr;
    p.k_stride_vec = p.v_stride_vec = p.max_pos = p.num_slots = 0;
    p.eps = 0.f;
    if (fused) {
        p.k_new = (const uint4*)fused->k_new;
        p.v_new = (const uint4*)fused->v_new;
        p.q_w = (const uint4*)fused->q_w;
        p.k_w = (const uint4*)fused->k_w;
        p.cos_sin = (const float4*)fused->cos_sin;
        p.positions = fused->positions;
        p.slots = fused->slots;
        p.k_cache_w = (uint4*)const_cast<void*>(k_cache);
        p.v_cache_w = (uint4*)const_cast<void*>(v_cache);
        p.k_stride_vec = fused->k_stride_t / 8;
        p.v_stride_vec = fused->v_stride_t / 8;
        p.max_pos = fused->max_pos;
        p.num_slots = num_blocks * (int64_t)block_size;
        p.eps = fused->eps;
    }
    hipStream_t st = (hipStream_t)stream;
    bool single = false;
    if (p.nsplit > 1) {
        const int64_t need = kTicketRegion + (int64_t)B * H_q * p.nsplit * kPartStride * (int64_t)sizeof(float);
        NV_REQUIRE(workspace && workspace_bytes >= need, NV_ERR_WORKSPACE,
                   "nv_attn_decode_paged: workspace %lld bytes < required %lld",
                   (long long)workspace_bytes, (long long)need);
        p.part = reinterpret_cast<float*>(static_cast<char*>(workspace) + kTicketRegion);
        // In-launch combine (the item that stores a pair's last partial state merges them): only on request
        // (nv_tuning.decode_single_launch = 1).  Measured against the second launch on gfx950 it wins 4 % at
        // 1 sequence x 32 k keys x 8 kv heads and loses 7 - 40 % where a launch has few (sequence, kv head)
        // pairs or wave-sized items (the combining item's acquire and its serial walk over the partial
        // states end the launch later than a 16-block merge kernel does).  A caller's workspace may hold
        // anything, so its counters are zeroed on the stream ahead of the launch.
        const int64_t ticket_bytes = ((int64_t)B * H_kv * (int64_t)sizeof(int) + 15) / 16 * 16;
        // On the library's own scratch (counters known to be zero, no memset on the stream) the in-launch combine
        // is also the default for launches with at most 8 (sequence, kv head) pairs, the one case where it
        // measured ahead of the second launch (1 sequence x 32 k keys x 8 kv heads: 31.7 - 32.0 against 32.2 - 33.0 us).
        const int sl = tuning().decode_single_launch;
        // ... and on a caller's workspace whose counters the caller vouches for (NV_DECODE_WS_COUNTERS_ZERO).
        const bool counters_zero = library_workspace || (flags & NV_DECODE_WS_COUNTERS_ZERO) != 0;
        single = (sl == 1 || (sl < 0 && counters_zero && (int64_t)B * H_kv <= 8)) && ticket_bytes <= kTicketRegion &&
                 (!fused || plan.wgs);   // (the fused instantiations carry the in-launch combine for workgroup items only)
        if (single) {
            p.tickets = static_cast<int*>(workspace);
            if (!counters_zero) {
                const hipError_t e = hipMemsetAsync(workspace, 0, (size_t)ticket_bytes, st);
                NV_REQUIRE(e == hipSuccess, NV_ERR_LAUNCH, "nv_attn_decode_paged: hipMemsetAsync failed: %s",
                           hipGetErrorString(e));
            }
        }
    }
    const bool wgs = plan.wgs, lean = plan.lean, use_mfma = plan.mfma, mfma_deep = plan.mfma_deep;
    // workgroups per key-range group; an even count gets a rotation of one workgroup per group (see the kernels'
    // item mapping)
    const int64_t group_wgs = wgs ? (int64_t)B * H_kv : ((int64_t)B * H_kv + kDecWaves - 1) / kDecWaves;
    const int64_t grid64 = group_wgs * p.nsplit;
    NV_REQUIRE(grid64 * kDecWaves < (1LL << 31), NV_ERR_UNSUPPORTED, "nv_attn_decode_paged: launch too large");
    p.group_items = (int)(wgs ? group_wgs : group_wgs * kDecWaves);
    // ... when the host has said that the batch is uneven (a chunk of its choice): with equal sequences nothing
    // sits on one XCD for long, and every (sequence, head group) staying on one XCD measured 1 - 1.5 % faster
    // (8 x 32 k keys: 176.5 against 179.5 us; its block table and query rows stay in that XCD's L2)
    p.group_rot = (chunk_keys > 0 && p.nsplit > 1 && group_wgs % 2 == 0) ? (wgs ? 1 : kDecWaves) : 0;
    const unsigned grid = (unsigned)grid64;
    const bool aligned = block_size % 16 == 0 && p.chunk % 64 == 0;
    // the in-launch combine lives in instantiations of its own (TK): the default kernels carry none of it
#define NV_DEC_LAUNCH3(GG, AL, TK)                                                                  \
    do {                                                                                            \
        if (wgs) hipLaunchKernelGGL((attn_decode_kernel<GG, AL, 1, true, TK>), dim3(grid), dim3(kDecWaves * 64), 0, st, p);       \
        else if (lean) hipLaunchKernelGGL((attn_decode_kernel<GG, AL, 0, false, TK>), dim3(grid), dim3(kDecWaves * 64), 0, st, p); \
        else hipLaunchKernelGGL((attn_decode_kernel<GG, AL, 1, false, TK>), dim3(grid), dim3(kDecWaves * 64), 0, st, p);           \
    } while (0)
#define NV_DEC_LAUNCH2(GG, AL)                   \
    do {                                         \
        if (single) NV_DEC_LAUNCH3(GG, AL, true); \
        else NV_DEC_LAUNCH3(GG, AL, false);      \
    } while (0)
#define NV_DEC_LAUNCH(GG)                    \
    do {                                     \
        if (aligned) NV_DEC_LAUNCH2(GG, true); \
        else NV_DEC_LAUNCH2(GG, false);      \
    } while (0)
#define NV_DEC_LAUNCH_M3(GG, AL, TK)                                                                \
    do {                                                                                            \
        if (wgs) hipLaunchKernelGGL((attn_decode_mfma_kernel<GG, AL, 1, true, TK>), dim3(grid), dim3(kDecWaves * 64), 0, st, p);            \
        else if (mfma_deep) hipLaunchKernelGGL((attn_decode_mfma_kernel<GG, AL, 1, false, TK>), dim3(grid), dim3(kDecWaves * 64), 0, st, p); \
        else hipLaunchKernelGGL((attn_decode_mfma_kernel<GG, AL, 0, false, TK>), dim3(grid), dim3(kDecWaves * 64), 0, st, p);               \
    } while (0)
#define NV_DEC_LAUNCH_M2(GG, AL)                   \
    do {                                           \
        if (single) NV_DEC_LAUNCH_M3(GG, AL, true); \
        else NV_DEC_LAUNCH_M3(GG, AL, false);      \
    } while (0)
#define NV_DEC_LAUNCH_M(GG)                      \
    do {                                         \
        if (aligned) NV_DEC_LAUNCH_M2(GG, true); \
        else NV_DEC_LAUNCH_M2(GG, false);        \
    } while (0)
#define NV_DEC_LAUNCH_F(GG)                                                                          \
    do {                                                                                            \
        if (wgs && single) hipLaunchKernelGGL((attn_decode_kernel<GG, true, 1, true, true, true>), dim3(grid), dim3(kDecWaves * 64), 0, st, p); \
        else if (wgs) hipLaunchKernelGGL((attn_decode_kernel<GG, true, 1, true, false, true>), dim3(grid), dim3(kDecWaves * 64), 0, st, p);   \
        else if (lean) hipLaunchKernelGGL((attn_decode_kernel<GG, true, 0, false, false, true>), dim3(grid), dim3(kDecWaves * 64), 0, st, p); \
        else hipLaunchKernelGGL((attn_decode_kernel<GG, true, 1, false, false, true>), dim3(grid), dim3(kDecWaves * 64), 0, st, p);           \
    } while (0)
#define NV_DEC_LAUNCH_MF(GG)                                                                         \
    do {                                                                                            \
        if (wgs && single) hipLaunchKernelGGL((attn_decode_mfma_kernel<GG, true, 1, true, true, true>), dim3(grid), dim3(kDecWaves * 64), 0, st, p); \
        else if (wgs) hipLaunchKernelGGL((attn_decode_mfma_kernel<GG, true, 1, true, false, true>), dim3(grid), dim3(kDecWaves * 64), 0, st, p);   \
        else if (mfma_deep) hipLaunchKernelGGL((attn_decode_mfma_kernel<GG, true, 1, false, false, true>), dim3(grid), dim3(kDecWaves * 64), 0, st, p); \
        else hipLaunchKernelGGL((attn_decode_mfma_kernel<GG, true, 0, false, false, true>), dim3(grid), dim3(kDecWaves * 64), 0, st, p);               \
    } while (0)
    if (fused) {
        NV_REQUIRE(aligned && (G <= 2 || use_mfma), NV_ERR_UNSUPPORTED,
                   "nv_attn_decode_paged_fused: block_size %d, H_q/H_kv = %d (supported: block sizes that are multiples "
                   "of 16; more than two query heads per kv head on the MFMA score path only); run "
                   "nv_qknorm_rope_store and nv_attn_decode_paged", block_size, G);
        if (G == 1) NV_DEC_LAUNCH_F(1);
        else if (G == 2) NV_DEC_LAUNCH_F(2);
        else if (G <= 4) NV_DEC_LAUNCH_MF(4);
        else NV_DEC_LAUNCH_MF(8);
    } else
    // G = 3, 5, 6, 7 run the MFMA-score kernel laid out for the next of 4 / 8 heads
    switch (G) {
        case 1: NV_DEC_LAUNCH(1); break;
        case 2: NV_DEC_LAUNCH(2); break;
        case 3: NV_DEC_LAUNCH_M(4); break;
        case 4:
            if (use_mfma) NV_DEC_LAUNCH_M(4);
            else NV_DEC_LAUNCH(4);
            break;
        case 8:
            if (use_mfma) NV_DEC_LAUNCH_M(8);
            else NV_DEC_LAUNCH(8);
            break;
        default: NV_DEC_LAUNCH_M(8); break;
    }
#undef NV_DEC_LAUNCH_F
#undef NV_DEC_LAUNCH_MF
#undef NV_DEC_LAUNCH3
#undef NV_DEC_LAUNCH2
#undef NV_DEC_LAUNCH
#undef NV_DEC_LAUNCH_M3
#undef NV_DEC_LAUNCH_M2
#undef NV_DEC_LAUNCH_M
    int rc = check_launch("nv_attn_decode_paged");
    if (rc != NV_OK) return rc;
    if (p.nsplit > 1 && !single) {
        if (p.nsplit > 16) {
            hipLaunchKernelGGL(attn_decode_merge_kernel<8>, dim3((unsigned)((int64_t)B * H_q)), dim3(256), 0, st, p, H_q);
        } else {
            const unsigned mgrid = (unsigned)(((int64_t)B * H_q + 7) / 8);
            hipLaunchKernelGGL(attn_decode_merge_kernel<1>, dim3(mgrid), dim3(256), 0, st, p, H_q);
        }
        rc = check_launch("nv_attn_decode_paged(merge)");
    }
    return rc;
}

// ---- the allocation-free entry point's scratch ---------------------------------------------
// One array per GPU, part of the code object, alive until the process ends.  In host code the
// array's name is only a handle: its device address on the current GPU comes from
// hipGetSymbolAddress and is kept per device.
constexpr int64_t kStaticScratchBytes = 32ll << 20;
constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
static void* g_scratch_ptr[kMaxDevices] = {};

// Device address of `scratch` on the current device (looked up once per device).
static int static_scratch_ptr(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    NV_REQUIRE(e == hipSuccess && dev >= 0 && dev < kMaxDevices, NV_ERR_LAUNCH,
               "nv_attn_decode_paged_static: hipGetDevice failed or device index %d out of range", dev);
    void* ptr = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!ptr) {
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        NV_REQUIRE(e == hipSuccess && ptr, NV_ERR_LAUNCH,
                   "nv_attn_decode_paged_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
        __atomic_store_n(&g_scratch_ptr[dev], ptr, __ATOMIC_RELEASE);
    }
    *out = ptr;
    return NV_OK;
}
}  // namespace nv

extern "C" int nv_attn_decode_paged(const void* q, const void* k_cache, const void* v_cache, void* o,
                                    const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                    const int32_t* block_tables, int B, int max_blocks,
                                    int max_seqlen_k, int H_q, int H_kv, int D, int64_t num_blocks,
                                    int block_size, float scale, int64_t q_stride_t,
                                    int64_t o_stride_t, void* workspace, int64_t workspace_bytes,
                                    void* stream) {
    return nv::decode_launch(q, k_cache, v_cache, o, cu_seqlens_q, cu_seqlens_k, block_tables, B, max_blocks,
                             max_seqlen_k, H_q, H_kv, D, num_blocks, block_size, scale, q_stride_t, o_stride_t,
                             workspace, workspace_bytes, false, stream);
}

extern "C" int nv_attn_decode_paged_chunked(const void* q, const void* k_cache, const void* v_cache, void* o,
                                            const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                            const int32_t* block_tables, int B, int max_blocks,
                                            int max_seqlen_k, int H_q, int H_kv, int D, int64_t num_blocks,
                                            int block_size, float scale, int64_t q_stride_t,
                                            int64_t o_stride_t, void* workspace, int64_t workspace_bytes,
                                            int chunk_keys, int flags, void* stream) {
    return nv::decode_launch(q, k_cache, v_cache, o, cu_seqlens_q, cu_seqlens_k, block_tables, B, max_blocks,
                             max_seqlen_k, H_q, H_kv, D, num_blocks, block_size, scale, q_stride_t, o_stride_t,
                             workspace, workspace_bytes, false, stream, nullptr, chunk_keys, flags);
}

extern "C" int nv_attn_decode_paged_fused_chunked(const void* q, const void* k, const void* v, const void* q_w,
                                                  const void* k_w, const float* cos_sin, const int64_t* positions,
                                                  const int32_t* slot_mapping, void* k_cache, void* v_cache, void* o,
                                                  const int32_t* cu_seqlens_k, const int32_t* block_tables, int B,
                                                  int max_blocks, int max_seqlen_k, int H_q, int H_kv, int D,
                                                  int64_t num_blocks, int block_size, int64_t max_pos, float eps,
                                                  float scale, int64_t q_stride_t, int64_t k_stride_t,
                                                  int64_t v_stride_t, int64_t o_stride_t, void* workspace,
                                                  int64_t workspace_bytes, int chunk_keys, int flags, void* stream) {
    using namespace nv;
    NV_REQUIRE(B >= 0 && H_kv > 0 && max_pos > 0, NV_ERR_INVALID_ARG,
               "nv_attn_decode_paged_fused: bad sizes B=%d H_kv=%d max_pos=%lld", B, H_kv, (long long)max_pos);
    if (B == 0) return NV_OK;
    NV_REQUIRE(k && v && cos_sin && positions && slot_mapping, NV_ERR_INVALID_ARG,
               "nv_attn_decode_paged_fused: null pointer");
    NV_REQUIRE((q_w == nullptr) == (k_w == nullptr), NV_ERR_INVALID_ARG,
               "nv_attn_decode_paged_fused: q_w and k_w must both be given or both be NULL");
    NV_REQUIRE(k_stride_t % 8 == 0 && v_stride_t % 8 == 0 && k_stride_t >= (int64_t)H_kv * D &&
                   v_stride_t >= (int64_t)H_kv * D,
               NV_ERR_INVALID_ARG, "nv_attn_decode_paged_fused: bad token strides");
    NV_REQUIRE(((uintptr_t)k | (uintptr_t)v | (uintptr_t)q_w | (uintptr_t)k_w | (uintptr_t)cos_sin) % 16 == 0,
               NV_ERR_INVALID_ARG, "nv_attn_decode_paged_fused: pointers must be 16-byte aligned");
    FusedDecodeIn f;
    f.k_new = k;
    f.v_new = v;
    f.q_w = q_w;
    f.k_w = k_w;
    f.cos_sin = cos_sin;
    f.positions = positions;
    f.slots = slot_mapping;
    f.k_stride_t = k_stride_t;
    f.v_stride_t = v_stride_t;
    f.max_pos = max_pos;
    f.eps = eps;
    return decode_launch(q, k_cache, v_cache, o, nullptr, cu_seqlens_k, block_tables, B, max_blocks, max_seqlen_k,
                         H_q, H_kv, D, num_blocks, block_size, scale, q_stride_t, o_stride_t, workspace,
                         workspace_bytes, false, stream, &f, chunk_keys, flags);
}

extern "C" int nv_attn_decode_paged_fused(const void* q, const void* k, const void* v, const void* q_w, const void* k_w,
                                          const float* cos_sin, const int64_t* positions,
                                          const int32_t* slot_mapping, void* k_cache, void* v_cache, void* o,
                                          const int32_t* cu_seqlens_k, const int32_t* block_tables, int B,
                                          int max_blocks, int max_seqlen_k, int H_q, int H_kv, int D,
                                          int64_t num_blocks, int block_size, int64_t max_pos, float eps, float scale,
                                          int64_t q_stride_t, int64_t k_stride_t, int64_t v_stride_t,
                                          int64_t o_stride_t, void* workspace, int64_t workspace_bytes,
                                          void* stream) {
    return nv_attn_decode_paged_fused_chunked(q, k, v, q_w, k_w, cos_sin, positions, slot_mapping, k_cache, v_cache, o,
                                              cu_seqlens_k, block_tables, B, max_blocks, max_seqlen_k, H_q, H_kv, D,
                                              num_blocks, block_size, max_pos, eps, scale, q_stride_t, k_stride_t,
                                              v_stride_t, o_stride_t, workspace, workspace_bytes, 0, 0, stream);
}

extern "C" int64_t nv_attn_decode_static_scratch_bytes(void) {
    void* ptr = nullptr;
    if (nv::static_scratch_ptr(&ptr) != NV_OK) return -1;
    return nv::kStaticScratchBytes;
}

extern "C" int nv_attn_decode_paged_static(const void* q, const void* k_cache, const void* v_cache, void* o,
                                           const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                           const int32_t* block_tables, int B, int max_blocks,
                                           int max_seqlen_k, int H_q, int H_kv, int D, int64_t num_blocks,
                                           int block_size, float scale, int64_t q_stride_t,
                                           int64_t o_stride_t, void* stream) {
    using namespace nv;
    const int64_t need = nv_attn_decode_workspace_bytes(B, H_q, H_kv, D, max_seqlen_k);
    NV_REQUIRE(need <= kStaticScratchBytes, NV_ERR_WORKSPACE,
               "nv_attn_decode_paged_static: this launch needs %lld bytes of split-key partial states, the "
               "built-in scratch holds %lld; use nv_attn_decode_paged with a caller workspace",
               (long long)need, (long long)kStaticScratchBytes);
    // The scratch address is resolved for the CURRENT device: a stream of another device would make the kernels
    // of that device write through this device's address.
    if (stream) {
        int cur = -1, sdev = -1;
        if (hipGetDevice(&cur) == hipSuccess && hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess)
            NV_REQUIRE(cur == sdev, NV_ERR_INVALID_ARG,
                       "nv_attn_decode_paged_static: the stream belongs to device %d but device %d is current; make "
                       "the stream's device current before the call", sdev, cur);
    }
    void* ptr = nullptr;
    const int rc = static_scratch_ptr(&ptr);
    if (rc != NV_OK) return rc;
    // The scratch's first kTicketRegion bytes only ever hold the arrival counters of the in-launch combine:
    // zero when the code object is loaded, and put back to zero by every launch that used them.  Every
    // partial state a combine reads is written by this call's attention kernel first.
    return decode_launch(q, k_cache, v_cache, o, cu_seqlens_q, cu_seqlens_k, block_tables, B, max_blocks,
                         max_seqlen_k, H_q, H_kv, D, num_blocks, block_size, scale, q_stride_t, o_stride_t, ptr,
                         kStaticScratchBytes, true, stream);
}
