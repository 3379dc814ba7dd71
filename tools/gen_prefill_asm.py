#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 main loop of the prefill attention kernel.

    python tools/gen_prefill_asm.py            # writes nano-vllm_amd/csrc/attn_prefill_loop.inc

The output is ONE inline-asm body (a C++ raw string) used by csrc/attn_prefill_asm.hip.  A wave owns 64
query rows (two 32-row blocks qb) of one query head and the whole 512-entry register file:

  AGPR  a[0:127]    O^T accumulators   O(qb, d, i)   = 64 qb + 16 d + i        (d = 32-dim block)
        a[128:191]  Q fragments        Q(qb, s, w)   = 128 + 32 qb + 4 s + w   (s = 16-dim k-step)
        a[192:255]  K fragments        K(kb, s, w)   = 192 + 32 kb + 4 s + w   (kb = 32-key block)
  VGPR  v[0:127]    S^T, two tiles     S(buf, kb, qb, i) = 64 buf + 32 kb + 16 qb + i
        v[128:191]  V^T fragments      V(d, ks, w)   = 128 + 16 d + 4 ks + w   (ks = 16-key k-step)
        v[192:223]  P^T fragments      P(ks, qb, w)  = 192 + 8 ks + 4 qb + w
        v[224:255]  softmax state, lane-constant addresses, temporaries

Tile j (64 keys) is processed by a two-phase software pipeline; iteration j of the loop is
  phase 1: 32 MFMAs  S(j) = K(j) Q^T     beside  exp / row sums / bf16 packing of the second half of S(j-1),
                                                 transposed LDS reads of V(j-1), LDS-DMA of K(j+2), V(j)
  phase 2: 32 MFMAs  O += V(j-1)^T P(j-1) beside  row maxima of S(j), exp of its first half, LDS reads of
                                                 K(j+1)
with at most ~5 single-issue instructions behind each MFMA (one wave per SIMD hides about that many).
One workgroup barrier per tile; K/V tiles arrive by LDS-DMA one iteration before they are read.
Masked tiles (causal diagonal, end of the sequence) start their S accumulators from a 0 / -inf tile
instead of 0; the accumulator rescale (running maximum raised by more than 2^8) is a rare side path.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "nano-vllm_amd", "csrc", "attn_prefill_loop.inc")

# ---- register map ------------------------------------------------------------------------------
def S(buf, kb, qb, i=0): return 64 * buf + 32 * kb + 16 * qb + i
def Vf(d, ks, w=0): return 128 + 16 * d + 4 * ks + w
def P(ks, qb, w=0): return 192 + 8 * ks + 4 * qb + w
def O(qb, d, i=0): return 64 * qb + 16 * d + i
def Q(qb, s, w=0): return 128 + 32 * qb + 4 * s + w
def K(kb, s, w=0): return 192 + 32 * kb + 4 * s + w

M = [224, 225]      # running maximum per 32-row block (log2 domain, scale folded in)
L = [226, 227]      # this lane's share of the running denominator
AL = [228, 229]     # pending accumulator rescale factor
TM = [230, 231]     # tile maximum (raw scores)
KB = [232, 233]     # LDS read bases of the K fragments (k-step even / odd)
VB = [234, 235]     # LDS read bases of the V^T fragments (rows +0 / +8)
DOFFK = [236, 237, 238, 239]   # per-lane source offsets of this wave's four DMA pieces, K tiles
DOFFV = [240, 241, 242, 243]   # the same for V tiles
QPOS = [244, 245]   # last visible key of the lane's query row, minus 4 * (lane >> 5)
ZERO, NINF, PADDR, STG = 246, 247, 248, 249
T = [250, 251, 252, 253, 254, 255]   # temporaries; v[252:253] / v[254:255] carry tile-table entries across iterations

# scalar registers (all listed as clobbers)
S_KBASE, S_VBASE = 36, 38      # 64-bit cache bases (kv head folded in)
S_NT, S_J, S_C, S_TAIL, S_MASKFROM, S_MASKED, S_RESC, S_PIECE, S_STG, S_TBL = 40, 41, 42, 43, 44, 45, 46, 47, 48, 49
S_T0 = 50                      # s50..s57 temporaries
S_CMP = 60                     # s[60:67] compare results

# LDS layout (bytes)
SLOT = 32768                   # one K tile (16 KiB) + one V tile (16 KiB)
V_OFF = 16384
PARAM_OFF = 16384              # per-wave staging area: [16 Q fragments x 1 KiB][16 lane parameters x 256 B]
MFMA = "v_mfma_f32_32x32x16_bf16"
NEG_BIG = "0xf149f2ca"         # -1.0e30f
EIGHT = "0x41000000"           # 8.0f


class Out:
    def __init__(self):
        self.lines: list[str] = []

    def e(self, s: str):
        self.lines.append("  " + s)

    def label(self, s: str):
        self.lines.append(s + ":")

    def c(self, s: str):
        self.lines.append("  ; " + s)


def rng(kind, base, n):
    return f"{kind}[{base}:{base + n - 1}]"


# ---- instruction streams -------------------------------------------------------------------------
def qk_mfmas(buf, masked_label_prefix):
    """32 MFMAs S^T[kb][qb] = sum_s K[kb][s] Q[qb][s]; chain starts take C = 0, or C = the preset mask tile
    through an out-of-line copy when the tile is masked.  Returns (list of (text, out_of_line_text))."""
    ms = []
    n = 0
    for kb in range(2):
        for qb in range(2):
            for s in range(8):
                d = rng("v", S(buf, kb, qb), 16)
                a = rng("a", K(kb, s), 4)
                b = rng("a", Q(qb, s), 4)
                if s == 0:
                    ms.append(("start", d, a, b, f"{masked_label_prefix}_{n}"))
                    n += 1
                else:
                    ms.append(("acc", d, a, b, None))
    return ms


def pv_mfmas():
    ms = []
    for ks in range(4):
        for d in range(4):
            for qb in range(2):
                acc = rng("a", O(qb, d), 16)
                ms.append(f"{MFMA} {acc}, {rng('v', Vf(d, ks), 4)}, {rng('v', P(ks, qb), 4)}, {acc}")
    return ms


def v_reads(slot):
    """32 transposed reads of a V tile into the V^T fragment registers, k-step major."""
    out = []
    for ks in range(4):
        for d in range(4):
            off = slot * SLOT + V_OFF + 4096 * ks + 512 * d
            out.append(f"ds_read_b64_tr_b16 {rng('v', Vf(d, ks, 0), 2)}, v{VB[0]} offset:{off}")
            out.append(f"ds_read_b64_tr_b16 {rng('v', Vf(d, ks, 2), 2)}, v{VB[1]} offset:{off}")
    return out


def k_reads(slot):
    out = []
    for kb in range(2):
        for s in range(8):
            off = slot * SLOT + 8192 * kb + 512 * (s >> 1)
            out.append(f"ds_read_b128 {rng('a', K(kb, s), 4)}, v{KB[s & 1]} offset:{off}")
    return out


def exp_stream(buf, kb, with_cvt=True):
    """exp2(S * c - m) in place, row sums, packing to bf16 pairs for the S registers of key block kb.
    Pattern per two values: fma a, exp a, fma b, exp b, add a, add b, cvt(a, b)."""
    out = []
    for qb in range(2):
        for i in range(0, 16, 2):
            ra, rb = S(buf, kb, qb, i), S(buf, kb, qb, i + 1)
            out.append(f"v_fma_f32 v{T[0]}, v{ra}, s{S_C}, -v{M[qb]}")
            out.append(f"v_exp_f32 v{ra}, v{T[0]}")
            out.append(f"v_fma_f32 v{T[1]}, v{rb}, s{S_C}, -v{M[qb]}")
            out.append(f"v_exp_f32 v{rb}, v{T[1]}")
            out.append(f"v_add_f32 v{L[qb]}, v{L[qb]}, v{ra}")
            out.append(f"v_add_f32 v{L[qb]}, v{L[qb]}, v{rb}")
            if with_cvt:
                ks = 2 * kb + (i >> 3)
                w = (i & 7) >> 1
                out.append(("cvt", f"v_cvt_pk_bf16_f32 v{P(ks, qb, w)}, v{ra}, v{rb}"))
    return out


def max_stream(buf):
    """Row maxima of S(buf) for both query blocks (raw scores): kb = 0 registers first."""
    out = []
    for qb in range(2):
        regs = [S(buf, 0, qb, i) for i in range(16)] + [S(buf, 1, qb, i) for i in range(16)]
        t = TM[qb]
        out.append(f"v_max3_f32 v{t}, v{regs[0]}, v{regs[1]}, v{regs[2]}")
        for k in range(3, 31, 2):
            out.append(f"v_max3_f32 v{t}, v{t}, v{regs[k]}, v{regs[k + 1]}")
        out.append(f"v_max_f32 v{t}, v{t}, v{regs[31]}")
    # exchange with the lane that holds the other half of the row (l ^ 32)
    out.append(f"v_mov_b32 v{T[0]}, v{TM[0]}")
    out.append(f"v_mov_b32 v{T[1]}, v{TM[1]}")
    out.append("s_nop 1")
    out.append(f"v_permlane32_swap_b32 v{TM[0]}, v{T[0]}")
    out.append(f"v_permlane32_swap_b32 v{TM[1]}, v{T[1]}")
    out.append(f"v_max_f32 v{TM[0]}, v{TM[0]}, v{T[0]}")
    out.append(f"v_max_f32 v{TM[1]}, v{TM[1]}, v{T[1]}")
    return out


STATS: dict = {}


def interleave(o: Out, mfmas, fillers, tag):
    """One MFMA, then its share of the fillers.  A filler is a string or (min_gap, string): the
    instruction may not be placed before MFMA number min_gap has been issued."""
    n_m = len(mfmas)
    items = [(f if isinstance(f, tuple) else (0, f)) for f in fillers]
    total = len(items)
    pos = 0
    pending: list[tuple[int, str]] = []
    STATS[tag] = (len(mfmas), total)
    for g, m in enumerate(mfmas):
        emit_mfma(o, m)
        want = (g + 1) * total // n_m - g * total // n_m
        placed = 0
        # delayed fillers whose gap has come go first
        keep = []
        for mg, txt in pending:
            if mg <= g and placed < want + 2:
                o.e(txt)
                placed += 1
            else:
                keep.append((mg, txt))
        pending = keep
        while placed < want and pos < total:
            mg, txt = items[pos]
            pos += 1
            if mg > g:
                pending.append((mg, txt))
                continue
            o.e(txt)
            placed += 1
    for mg, txt in pending:
        o.e(txt)
    while pos < total:
        o.e(items[pos][1])
        pos += 1


_ool: list[list[str]] = []     # out-of-line blocks, appended after the main code


def emit_mfma(o: Out, m):
    if isinstance(m, str):
        o.e(m)
        return
    kind, d, a, b, lab = m
    if kind == "acc":
        o.e(f"{MFMA} {d}, {a}, {b}, {d}")
        return
    # chain start: C = 0, or C = the mask tile already sitting in the accumulator registers
    o.e(f"s_cmp_lg_u32 s{S_MASKED}, 0")
    o.e(f"s_cbranch_scc1 {lab}")
    o.e(f"{MFMA} {d}, {a}, {b}, 0")
    o.label(f"{lab}_ret")
    _ool.append([f"{lab}:", f"  {MFMA} {d}, {a}, {b}, {d}", f"  s_branch {lab}_ret"])


# ---- building blocks -------------------------------------------------------------------------------
def mask_setup(o: Out, buf, tag):
    """s_masked = (j >= mask_from); a masked tile presets S(buf) with 0 (visible) / -inf (hidden)."""
    o.e(f"s_cmp_ge_u32 s{S_J}, s{S_MASKFROM}")
    o.e(f"s_cselect_b32 s{S_MASKED}, 1, 0")
    o.e(f"s_cbranch_scc1 .Lpf_mask_{tag}")
    o.label(f".Lpf_mask_{tag}_ret")
    blk = [f".Lpf_mask_{tag}:", f"  s_lshl_b32 s{S_T0}, s{S_J}, 6"]
    for qb in range(2):
        blk.append(f"  v_subrev_u32 v{T[qb]}, s{S_T0}, v{QPOS[qb]}")
    blk.append("  s_nop 1")
    for qb in range(2):
        for kb in range(2):
            for i0 in range(0, 16, 4):
                for k in range(4):
                    i = i0 + k
                    key = kb * 32 + (i & 3) + 8 * (i >> 2)
                    blk.append(f"  v_cmp_ge_i32 s[{S_CMP + 2 * k}:{S_CMP + 2 * k + 1}], v{T[qb]}, {key}")
                for k in range(4):
                    blk.append(f"  v_cndmask_b32 v{S(buf, kb, qb, i0 + k)}, v{NINF}, v{ZERO}, "
                               f"s[{S_CMP + 2 * k}:{S_CMP + 2 * k + 1}]")
    blk.append("  s_nop 3")
    blk.append(f"  s_branch .Lpf_mask_{tag}_ret")
    _ool.append(blk)


def dma_pieces(kind, slot, offregs):
    """Four 1 KiB LDS-DMA pieces of this wave for a K or V tile whose 64-bit source base (minus 4096) sits
    in s[54:55].  M0 holds the LDS address of the wave's first piece; piece i goes through the immediate
    offset i * 1024, which the hardware adds to the LDS address AND to the global address: the per-lane
    source offsets are stored as offset + 4096 - 1024 i (the 4096 keeps them non-negative)."""
    dst = slot * SLOT + (V_OFF if kind == "V" else 0)
    out = [f"s_add_u32 m0, s{S_PIECE}, {dst}", "s_nop 0"]
    for i in range(4):
        out.append(f"global_load_lds_dwordx4 v{offregs[i]}, s[54:55] offset:{i * 1024}")
    return out


def tail_fix(o: Out, tile_sreg, offregs, tag):
    """The tile that holds the end of the sequence re-reads its last valid row for the rows behind it
    (finite data instead of whatever the cache block holds): switch this wave's source offsets to the
    clamped set (lane parameters 8..11) when that tile is about to be fetched."""
    o.e(f"s_cmp_eq_u32 s{tile_sreg}, s{S_TAIL}")
    o.e(f"s_cbranch_scc1 .Lpf_tail_{tag}")
    o.label(f".Lpf_tail_{tag}_ret")
    blk = [f".Lpf_tail_{tag}:"]
    for i in range(4):
        blk.append(f"  ds_read_b32 v{offregs[i]}, v{PADDR} offset:{PARAM_OFF + (8 + i) * 256}")
    blk += ["  s_waitcnt lgkmcnt(0)", f"  s_branch .Lpf_tail_{tag}_ret"]
    _ool.append(blk)


def table_lookup(o_list, tile_off_bytes_K, tile_off_bytes_V):
    """ds_read the byte offsets of two tiles (relative to the entry of tile j) into v[252:253] (K) and
    v[254:255] (V); T[0] is the address temp."""
    o_list.append(f"v_mov_b32 v{T[0]}, s{S_J}")
    o_list.append(f"v_lshl_add_u32 v{T[0]}, v{T[0]}, 3, s{S_TBL}")
    o_list.append(f"ds_read_b64 v[252:253], v{T[0]} offset:{tile_off_bytes_K}")
    o_list.append(f"ds_read_b64 v[254:255], v{T[0]} offset:{tile_off_bytes_V}")


def slow_max(o: Out, tag):
    """Decision of the deferred rescale: any row whose tile maximum exceeds the running one by more than
    2^8 sends the wave through the side path that raises the maxima and records the factors."""
    o.e(f"v_fma_f32 v{T[0]}, v{TM[0]}, s{S_C}, -v{M[0]}")
    o.e(f"v_fma_f32 v{T[1]}, v{TM[1]}, s{S_C}, -v{M[1]}")
    o.e(f"v_max_f32 v{T[0]}, v{T[0]}, v{T[1]}")
    o.e(f"v_cmp_lt_f32 vcc, {EIGHT}, v{T[0]}")
    o.e(f"s_cbranch_vccnz .Lpf_slow_{tag}")
    o.label(f".Lpf_slow_{tag}_ret")
    blk = [f".Lpf_slow_{tag}:"]
    for qb in range(2):
        blk += [f"  v_mul_f32 v{T[0]}, s{S_C}, v{TM[qb]}",
                f"  v_max_f32 v{T[0]}, v{M[qb]}, v{T[0]}",
                f"  v_sub_f32 v{T[1]}, v{M[qb]}, v{T[0]}",
                f"  v_exp_f32 v{AL[qb]}, v{T[1]}",
                f"  v_mov_b32 v{M[qb]}, v{T[0]}",
                "  s_nop 0",
                f"  v_mul_f32 v{L[qb]}, v{L[qb]}, v{AL[qb]}"]
    blk += [f"  s_mov_b32 s{S_RESC}, 1", f"  s_branch .Lpf_slow_{tag}_ret"]
    _ool.append(blk)


def rescale_check(o: Out, tag):
    o.e(f"s_cmp_lg_u32 s{S_RESC}, 0")
    o.e(f"s_cbranch_scc1 .Lpf_resc_{tag}")
    o.label(f".Lpf_resc_{tag}_ret")
    blk = [f".Lpf_resc_{tag}:", "  s_nop 15", "  s_nop 15"]
    tmp = [T[0], T[1], TM[0], TM[1]]     # v[252:255] carry the next tiles' table entries at this point
    for qb in range(2):
        for d in range(4):
            for i0 in range(0, 16, 4):
                for k in range(4):
                    blk.append(f"  v_accvgpr_read_b32 v{tmp[k]}, a{O(qb, d, i0 + k)}")
                blk.append("  s_nop 0")
                for k in range(4):
                    blk.append(f"  v_mul_f32 v{tmp[k]}, v{tmp[k]}, v{AL[qb]}")
                for k in range(4):
                    blk.append(f"  v_accvgpr_write_b32 a{O(qb, d, i0 + k)}, v{tmp[k]}")
    blk += [f"  s_mov_b32 s{S_RESC}, 0", "  s_nop 7", f"  s_branch .Lpf_resc_{tag}_ret"]
    _ool.append(blk)


def dma_block(o_list, p, tag_o: Out, tag):
    """Fillers that fetch K(j+2) and V(j) into slot p (tile-table entries already in v[252:255])."""
    f = o_list
    f.append(f"v_readfirstlane_b32 s{S_T0 + 2}, v252")
    f.append(f"v_readfirstlane_b32 s{S_T0 + 3}, v253")
    f.append(f"v_readfirstlane_b32 s{S_T0 + 6}, v254")
    f.append(f"v_readfirstlane_b32 s{S_T0 + 7}, v255")
    return f


# ---- sections ----------------------------------------------------------------------------------------
def gen_prologue(o: Out):
    o.c("---- set-up: operands to fixed registers, lane parameters and Q fragments from LDS ----")
    o.e(f"s_mov_b64 s[{S_KBASE}:{S_KBASE + 1}], %0")     # operands 0, 1: cache bases (kv head folded in) minus 4096
    o.e(f"s_mov_b64 s[{S_VBASE}:{S_VBASE + 1}], %1")
    o.e(f"s_mov_b32 s{S_NT}, %2")
    o.e(f"s_mov_b32 s{S_C}, %3")
    o.e(f"s_mov_b32 s{S_TAIL}, %4")
    o.e(f"s_mov_b32 s{S_MASKFROM}, %5")
    o.e(f"s_mov_b32 s{S_PIECE}, %6")
    o.e(f"s_mov_b32 s{S_STG}, %7")
    o.e(f"s_mov_b32 s{S_TBL}, %8")
    o.e(f"s_mov_b32 s{S_J}, 0")
    o.e(f"s_mov_b32 s{S_RESC}, 0")
    o.e(f"v_mbcnt_lo_u32_b32 v{T[0]}, -1, 0")
    o.e(f"v_mbcnt_hi_u32_b32 v{T[0]}, -1, v{T[0]}")
    o.e(f"v_lshlrev_b32 v{PADDR}, 2, v{T[0]}")
    o.e(f"v_add_u32 v{PADDR}, s{S_STG}, v{PADDR}")
    o.e(f"v_lshlrev_b32 v{T[1]}, 4, v{T[0]}")
    o.e(f"v_add_u32 v{T[1]}, s{S_STG}, v{T[1]}")
    plist = [KB[0], KB[1], VB[0], VB[1]] + DOFFK
    for pi, r in enumerate(plist):
        o.e(f"ds_read_b32 v{r}, v{PADDR} offset:{PARAM_OFF + pi * 256}")
    for i in range(4):
        o.e(f"ds_read_b32 v{DOFFV[i]}, v{PADDR} offset:{PARAM_OFF + (4 + i) * 256}")
    o.e(f"ds_read_b32 v{QPOS[0]}, v{PADDR} offset:{PARAM_OFF + 12 * 256}")
    o.e(f"ds_read_b32 v{QPOS[1]}, v{PADDR} offset:{PARAM_OFF + 13 * 256}")
    o.e(f"ds_read_b32 v{STG}, v{PADDR} offset:{PARAM_OFF + 14 * 256}")
    for qb in range(2):
        for s in range(8):
            o.e(f"ds_read_b128 {rng('a', Q(qb, s), 4)}, v{T[1]} offset:{(qb * 8 + s) * 1024}")
    o.e(f"v_mov_b32 v{ZERO}, 0")
    o.e(f"v_mov_b32 v{NINF}, 0xff800000")
    for qb in range(2):
        o.e(f"v_mov_b32 v{M[qb]}, {NEG_BIG}")
        o.e(f"v_mov_b32 v{L[qb]}, 0")
        o.e(f"v_mov_b32 v{AL[qb]}, 1.0")
    for i in range(128):
        o.e(f"v_accvgpr_write_b32 a{i}, 0")
    o.e("s_waitcnt lgkmcnt(0)")

    o.c("---- tile 0: fetch K0, V0 (slot 0) and K1 (slot 1), read K0 into registers, fetch K2 ----")

    def fetch(tile_const, kind, slot, offregs, tag, guard_nt):
        if guard_nt:
            o.e(f"s_cmp_le_u32 s{S_NT}, {tile_const}")
            o.e(f"s_cbranch_scc1 .Lpf_pro_skip_{tag}")
        o.e(f"s_mov_b32 s{S_T0}, {tile_const}")
        tail_fix(o, S_T0, offregs, f"pro_{tag}")
        o.e(f"v_mov_b32 v{T[0]}, s{S_TBL}")
        o.e(f"ds_read_b64 v[252:253], v{T[0]} offset:{8 * tile_const}")
        o.e("s_waitcnt lgkmcnt(0)")
        o.e("s_nop 1")
        o.e(f"v_readfirstlane_b32 s{S_T0 + 2}, v252")
        o.e(f"v_readfirstlane_b32 s{S_T0 + 3}, v253")
        base = S_VBASE if kind == "V" else S_KBASE
        o.e(f"s_add_u32 s54, s{base}, s{S_T0 + 2}")
        o.e(f"s_addc_u32 s55, s{base + 1}, s{S_T0 + 3}")
        for ln in dma_pieces(kind, slot, offregs):
            o.e(ln)
        if guard_nt:
            o.label(f".Lpf_pro_skip_{tag}")

    fetch(0, "K", 0, DOFFK, "k0", False)
    fetch(0, "V", 0, DOFFV, "v0", False)
    fetch(1, "K", 1, DOFFK, "k1", True)
    o.e("s_waitcnt vmcnt(0)")
    o.e("s_barrier")
    for ln in k_reads(0):
        o.e(ln)
    o.e("s_waitcnt lgkmcnt(0)")
    o.e("s_barrier")
    fetch(2, "K", 0, DOFFK, "k2", True)
    mask_setup(o, 0, "pro")
    for m in qk_mfmas(0, ".Lpf_mm_pro"):
        emit_mfma(o, m)
    o.e("s_nop 15")
    o.e("s_nop 7")
    o.c("tile 0: maxima define the first running maxima (no rescale: O = 0, l = 0)")
    for ln in max_stream(0):
        o.e(ln)
    for qb in range(2):
        o.e(f"v_mul_f32 v{T[0]}, s{S_C}, v{TM[qb]}")
        o.e(f"v_max_f32 v{M[qb]}, {NEG_BIG}, v{T[0]}")
    for ln in k_reads(1):
        o.e(ln)
    for x in exp_stream(0, 0):
        o.e(x[1] if isinstance(x, tuple) else x)
    lk: list[str] = []
    table_lookup(lk, 24, 8)          # next iteration j = 1 fetches K(3) and V(1); s_j is still 0 here
    for ln in lk:
        o.e(ln)
    o.e("s_waitcnt lgkmcnt(0)")
    o.e("s_waitcnt vmcnt(0)")
    o.e("s_barrier")
    o.e(f"s_mov_b32 s{S_J}, 1")
    o.e(f"s_cmp_le_u32 s{S_NT}, 1")
    o.e("s_cbranch_scc1 .Lpf_epi0")


def gen_body(o: Out, p):
    """Iteration j with j & 1 == p: S(j) -> buffer p; the pending tile j-1 sits in buffer 1-p."""
    q = 1 - p
    tag = f"b{p}"
    o.label(f".Lpf_body{p}")
    mask_setup(o, p, tag)
    # ---- phase 1 ----
    vr = v_reads(q)
    ex = exp_stream(q, 1)
    # V(j) and K(j+2) go to slot p.  The tile that holds the end of the sequence switches the wave's
    # source offsets to the clamped set (rare, checked here: branches stay out of the MFMA stream).
    # K(j+2) may lie behind the last tile: its table entry then repeats the last tile (a harmless fetch
    # into a slot nobody reads again).
    tail_fix(o, S_J, DOFFV, f"{tag}_v")
    o.e(f"s_add_u32 s{S_T0}, s{S_J}, 2")
    tail_fix(o, S_T0, DOFFK, f"{tag}_k")
    dma = [f"v_readfirstlane_b32 s{S_T0 + 6}, v254", f"v_readfirstlane_b32 s{S_T0 + 7}, v255",
           f"v_readfirstlane_b32 s{S_T0 + 2}, v252", f"v_readfirstlane_b32 s{S_T0 + 3}, v253",
           f"s_add_u32 s54, s{S_VBASE}, s{S_T0 + 6}", f"s_addc_u32 s55, s{S_VBASE + 1}, s{S_T0 + 7}"]
    dma += dma_pieces("V", p, DOFFV)
    dma += [f"s_add_u32 s54, s{S_KBASE}, s{S_T0 + 2}", f"s_addc_u32 s55, s{S_KBASE + 1}, s{S_T0 + 3}"]
    dma += dma_pieces("K", p, DOFFK)
    fill1: list = []
    # merge: fetches first, V reads early (needed at the start of phase 2), exponentials throughout
    vi = di = 0
    for k, x in enumerate(ex):
        fill1.append(x[1] if isinstance(x, tuple) else x)
        if k % 2 == 1 and di < len(dma):
            fill1.append(dma[di])
            di += 1
        if k % 3 == 1 and vi < len(vr):
            fill1.append(vr[vi])
            vi += 1
    fill1 += dma[di:] + vr[vi:]
    interleave(o, qk_mfmas(p, f".Lpf_mm_{tag}"), fill1, tag + "p1")
    o.e("s_waitcnt lgkmcnt(0)")
    # ---- phase 2 ----
    fill2: list = []
    mx = max_stream(p)
    kr = k_reads(q)
    # maxima first (kb = 0 registers lead; the last S accumulator was written by the final MFMA of
    # phase 1 and is read more than 16 instructions later), K reads beside them
    ki = 0
    for k, x in enumerate(mx):
        fill2.append(x)
        if k % 3 == 2 and ki < len(kr):
            fill2.append(kr[ki])
            ki += 1
    o2 = Out()
    interleave_phase2(o, p, q, fill2, kr[ki:], tag)


def interleave_phase2(o: Out, p, q, head, rest_kreads, tag):
    """Phase 2 is emitted in two parts because the rescale decision (a branch) sits between the maxima and
    the exponentials: MFMAs 0..n_head-1 carry the maxima, the rest carry exponentials / packing."""
    pv = pv_mfmas()
    n_head = 9
    interleave(o, pv[:n_head], head, tag + "p2a")
    slow_max(o, tag)
    tail: list = []
    ex = exp_stream(p, 0)
    ki = 0
    for k, x in enumerate(ex):
        if isinstance(x, tuple):
            # P fragments of k-steps 0, 1 are operands of the first 16 PV MFMAs of this phase
            tail.append((16 - n_head, x[1]))
        else:
            tail.append(x)
        if k % 6 == 5 and ki < len(rest_kreads):
            tail.append(rest_kreads[ki])
            ki += 1
    tail += rest_kreads[ki:]
    lk: list[str] = []
    table_lookup(lk, 24, 8)          # iteration j+1 fetches K(j+3) and V(j+1); s_j is still j here
    tail += lk
    interleave(o, pv[n_head:], tail, tag + "p2b")
    o.e("s_waitcnt lgkmcnt(0)")
    rescale_check(o, tag)
    o.e("s_waitcnt vmcnt(0)")
    o.e("s_barrier")
    o.e(f"s_add_u32 s{S_J}, s{S_J}, 1")
    o.e(f"s_cmp_ge_u32 s{S_J}, s{S_NT}")
    o.e(f"s_cbranch_scc1 .Lpf_epi{p}")
    if p == 0:
        o.e("s_branch .Lpf_body1")


def gen_epilogue(o: Out, qbuf):
    """After the last tile (which sits in buffer qbuf): finish its softmax, read its V, multiply."""
    o.label(f".Lpf_epi{qbuf}")
    for ln in v_reads(qbuf):
        o.e(ln)
    for x in exp_stream(qbuf, 1):
        o.e(x[1] if isinstance(x, tuple) else x)
    o.e("s_waitcnt lgkmcnt(0)")
    o.e("s_nop 3")
    for m in pv_mfmas():
        o.e(m)
    o.e("s_branch .Lpf_fin")


def gen_finish(o: Out):
    o.label(".Lpf_fin")
    o.c("denominators: add the share of the lane that holds the other half of each row; 1 / l (0 for rows without keys)")
    o.e("s_nop 15")
    o.e("s_nop 15")
    for qb in range(2):
        o.e(f"v_mov_b32 v{T[qb]}, v{L[qb]}")
    o.e("s_nop 1")
    for qb in range(2):
        o.e(f"v_permlane32_swap_b32 v{L[qb]}, v{T[qb]}")
    for qb in range(2):
        o.e(f"v_add_f32 v{L[qb]}, v{L[qb]}, v{T[qb]}")
    for qb in range(2):
        o.e(f"v_rcp_f32 v{AL[qb]}, v{L[qb]}")
        o.e(f"v_cmp_lt_f32 vcc, 0, v{L[qb]}")
        o.e("s_nop 1")
        o.e(f"v_cndmask_b32 v{AL[qb]}, 0, v{AL[qb]}, vcc")
    o.c("O^T -> bf16 rows in the wave's staging area: row qb*32 + (lane & 31), 272-byte row stride")
    for qb in range(2):
        for d in range(4):
            for g in range(4):
                for k in range(4):
                    o.e(f"v_accvgpr_read_b32 v{T[k]}, a{O(qb, d, 4 * g + k)}")
                o.e("s_nop 0")
                for k in range(4):
                    o.e(f"v_mul_f32 v{T[k]}, v{T[k]}, v{AL[qb]}")
                o.e(f"v_cvt_pk_bf16_f32 v{T[4]}, v{T[0]}, v{T[1]}")
                o.e(f"v_cvt_pk_bf16_f32 v{T[5]}, v{T[2]}, v{T[3]}")
                o.e("s_nop 0")
                o.e(f"ds_write_b64 v{STG}, v[{T[4]}:{T[5]}] offset:{qb * 32 * 272 + d * 64 + g * 16}")
    o.e("s_waitcnt lgkmcnt(0)")


def generate() -> str:
    o = Out()
    gen_prologue(o)
    gen_body(o, 1)
    gen_body(o, 0)
    gen_epilogue(o, 0)
    gen_epilogue(o, 1)
    gen_finish(o)
    o.e("s_branch .Lpf_end")
    for blk in _ool:
        o.lines += blk
    o.label(".Lpf_end")
    return "\n".join(o.lines) + "\n"


def clobbers() -> str:
    regs = [f"v{i}" for i in range(256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(36, 68)]
    regs += ["vcc", "scc", "memory"]
    return ", ".join(f'"{r}"' for r in regs)


def main():
    text = generate()
    with open(OUT, "w") as f:
        f.write("// Generated by tools/gen_prefill_asm.py -- do not edit; the schedule and the register map are\n"
                "// described there.  One string literal per instruction, spliced into a single asm statement.\n")
        for ln in text.splitlines():
            f.write('"' + ln.replace('"', "'") + '\\n"\n')
    with open(OUT.replace("_loop.inc", "_loop_clobbers.inc"), "w") as f:
        f.write("// Generated by tools/gen_prefill_asm.py: registers written by the asm body.\n")
        f.write(clobbers() + "\n")
    n_inst = sum(1 for ln in text.splitlines() if ln.startswith("  ") and not ln.strip().startswith(";"))
    print(f"wrote {OUT}: {n_inst} instructions")
    for tag, (nm, nf) in STATS.items():
        print(f"  {tag}: {nm} MFMAs, {nf} fillers = {nf / nm:.2f} per gap")


if __name__ == "__main__":
    sys.exit(main())
