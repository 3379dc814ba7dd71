#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 main loop of the prefill attention kernel.

    python tools/gen_prefill_asm.py            # writes nano-vllm_amd/csrc/attn_prefill_loop.inc

The output is ONE inline-asm body (a C++ raw string) used by csrc/attn_prefill_asm.hip.  A wave owns 64
query rows (two 32-row blocks qb) of one query head and the whole 512-entry register file:

  AGPR  a[0:127]    O^T accumulators   O(qb, d, i)   = 64 qb + 16 d + i        (d = 32-dim block)
        a[128:191]  Q fragments        Q(qb, s, w)   = 128 + 32 qb + 4 s + w   (s = 16-dim k-step)
        a[192:255]  K fragments        K(kb, s, w)   = 192 + 32 kb + 4 s + w   (kb = 32-key block)
  VGPR  v[0:127]    S^T, two tiles     S(buf, kb, qb, i) = 64 buf + 32 kb + 16 qb + i
        v[128:159]  V^T fragments      V(d, ks, w)   = 128 + 16 (ks & 1) + 4 d + w   (two of the four k-steps)
        v[160:191]  -m blocks          NMB(qb, i)    = 160 + 16 qb + i   (minus the running maximum, 16 copies)
        v[192:223]  P^T fragments      P(ks, qb, w)  = 192 + 8 ks + 4 qb + w
        v[224:255]  softmax state, lane-constant addresses, temporaries

Softmax scaling: the Q fragments are used as they are (a score is the fp32 sum of exact bf16 products, which is
what flash-attn's fp32 scaling sees), every S^T accumulator chain STARTS from the block that holds minus the
running maximum of its query rows (the MFMA's C operand, in raw score units), so an accumulator is already
q.k - m, and the probability is exp2(c * accumulator) with c = scale * log2(e): one v_mul_f32 per score in the exp
stream (PF_EXACT=1, the default).  A scale that is zero or negative reaches the loop as c = 1 or |c| and a factor
0 or -1 on the Q fragments, applied in a side path of the item's set-up that a positive scale never enters.
PF_EXACT=0 (timing only since the kernel's set-up passes the factors of the exact form) multiplies all of c into
the bf16 Q fragments instead and drops the multiply: 4 - 8 % faster, but the rounding of c * q to bf16 perturbs
every score by about 2^-9 of its magnitude, which shows on peaked softmax rows.  The running maximum is raised
(rarely: when a tile maximum exceeds it by more than 8 / c, i.e. 2^8 in the exponent) in a side path that also
shifts the tile's accumulators and the -m blocks.

Tile j (64 keys) is processed by a two-phase software pipeline; iteration j of the loop is
  phase 1: 32 MFMAs  S(j) = K(j) Q^T - m   beside  the rest of exp / row sums / bf16 packing of S(j-1),
                                                   transposed LDS reads of k-steps 0, 1 of V(j-1), LDS-DMA of
                                                   K(j+2), V(j)
  phase 2: 32 MFMAs  O += V(j-1)^T P(j-1)  beside  row maxima of S(j), the first part of its exp stream, LDS
                                                   reads of k-steps 2, 3 of V(j-1) (into the registers of
                                                   k-steps 0, 1 once those MFMAs have issued; counted waits)
                                                   and of K(j+1)
with 4 - 5 single-issue instructions behind each MFMA (one wave per SIMD hides about five).
One workgroup barrier per tile; K/V tiles arrive by LDS-DMA one iteration before they are read.
Masked tiles (causal diagonal, end of the sequence) start their S accumulators from a -m / -inf tile.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "nano-vllm_amd", "csrc", "attn_prefill_loop.inc")

# ---- register map ------------------------------------------------------------------------------
def S(buf, kb, qb, i=0): return 64 * buf + 32 * kb + 16 * qb + i
def Vf(d, ks, w=0): return 128 + 16 * (ks & 1) + 4 * d + w
def NMB(qb, i=0): return 160 + 16 * qb + i
def P(ks, qb, w=0): return 192 + 8 * ks + 4 * qb + w
def O(qb, d, i=0): return 64 * qb + 16 * d + i
def Q(qb, s, w=0): return 128 + 32 * qb + 4 * s + w
def K(kb, s, w=0): return 192 + 32 * kb + 4 * s + w

L = [226, 246]      # this lane's share of the running denominator: two partial sums per block, v[L:L+1]
TM = [230, 231]     # tile maximum (raw scores)
AL = TM             # pending accumulator rescale factor: written by the side path after it has consumed the maximum
KB = [232, 233]     # LDS read bases of the K fragments (k-step even / odd)
VB = [234, 235]     # LDS read bases of the V^T fragments (rows +0 / +8)
DOFFK = [236, 237, 238, 239]   # per-lane source offsets of this wave's four DMA pieces, K tiles
DOFFV = [240, 241, 242, 243]   # the same for V tiles
QPOS = [244, 245]   # last visible key of the lane's query row, minus 4 * (lane >> 5)
PADDR, NINF = 248, 249
T = [250, 251, 252, 253, 254, 255]   # temporaries; v[252:253] / v[254:255] carry tile-table entries across iterations

# scalar registers (all listed as clobbers)
S_KBASE, S_VBASE = 36, 38      # 64-bit cache bases (kv head folded in)
S_NT, S_J, S_C, S_TAIL, S_MASKFROM, S_MASKED, S_RESC, S_PIECE, S_STG, S_TBL = 40, 41, 42, 43, 44, 45, 46, 47, 48, 49
S_T0 = 50                      # s50..s57 temporaries
S_CMP = 60                     # s[60:67] compare results
S_DESC, S_NITEMS, S_ITEM = 68, 69, 70      # LDS address of the wave's item descriptors, item count, current item
S_MODE = 71                    # 0: bf16 output rows; 1: the unnormalised fp32 state of a key range (partial_state_store)
S_QSEQ, S_OSEQ = 72, 74        # 64-bit: row 0 of this wave's query head in q / in o (of the sequence)
S_QSTR, S_OSTR, S_NQ1 = 76, 77, 78        # token strides in bytes, n_q - 1
S_ITEMBASE, S_RW0, S_ROWS = 79, 80, 81    # per item: rw0 + shift, first row of the wave, rows that exist
S_SAVE = 82                    # s[82:83] saved exec
S_LAST = 84                    # index of the next item while an epilogue prefetches for it
S_R = 85                       # c = |scale * log2(e)| (fp32 bits; 1 for scale 0): the exp stream's multiplier
S_THR = 58                     # 8 / c: a tile maximum this far above the running maximum sends the wave to the side path

# LDS layout (bytes)
SLOT = 32768                   # one K tile (16 KiB) + one V tile (16 KiB)
V_OFF = 16384
PARAM_OFF = 16384              # per-wave staging area: [16 Q fragments x 1 KiB][16 lane parameters x 256 B]
FIRST_KV_PIECES = 12           # LDS-DMA instructions per wave of K(0), V(0), K(1), requested behind the first item's Q image
MFMA = "v_mfma_f32_32x32x16_bf16"
NEG_BIG = "0xf149f2ca"         # -1.0e30f
POS_BIG = "0x7149f2ca"         # +1.0e30f
EIGHT = "0x41000000"           # 8.0f


class Out:
    def __init__(self):
        self.lines: list[str] = []
        self.hot = False          # inside the loop bodies (ablations apply there only)

    def e(self, s: str):
        if ABLATE and self.hot and ablated(s):
            return
        self.lines.append("  " + s)

    def label(self, s: str):
        self.lines.append(s + ":")

    def c(self, s: str):
        self.lines.append("  ; " + s)


def rng(kind, base, n):
    return f"{kind}[{base}:{base + n - 1}]"


# ---- instruction streams -------------------------------------------------------------------------
def qk_mfmas(buf, masked_label_prefix, cinit=True, mode="branch"):
    """32 MFMAs S^T[kb][qb] = sum_s K[kb][s] Q[qb][s] (+ C); a chain starts from the -m block of its query
    rows (cinit) or from 0 (first tile of an item), or, when the tile is masked, from the preset mask tile
    through an out-of-line copy.  Returns a list of (kind, d, a, b, label, c)."""
    ms = []
    n = 0
    for kb in range(2):
        for qb in range(2):
            for s in range(8):
                d = rng("v", S(buf, kb, qb), 16)
                a = rng("a", K(kb, s), 4)
                b = rng("a", Q(qb, s), 4)
                if s == 0:
                    c = rng("v", NMB(qb), 16) if cinit else "0"
                    kind = {"branch": "start", "plain": "start_plain", "masked": "start_masked"}[mode]
                    ms.append((kind, d, a, b, f"{masked_label_prefix}_{n}", c))
                    n += 1
                else:
                    ms.append(("acc", d, a, b, None, None))
    return ms


def pv_mfmas():
    ms = []
    for ks in range(4):
        for d in range(4):
            for qb in range(2):
                acc = rng("a", O(qb, d), 16)
                ms.append(f"{MFMA} {acc}, {rng('v', Vf(d, ks), 4)}, {rng('v', P(ks, qb), 4)}, {acc}")
    return ms


def v_reads(slot, ks_list):
    """Transposed reads of the given k-steps of a V tile into the V^T fragment registers (k-steps 0 / 2 and
    1 / 3 share registers)."""
    out = []
    for ks in ks_list:
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


EXACT = os.environ.get("PF_EXACT", "1") != "0"   # fp32 score scaling (one multiply per score); 0: all of c in bf16 Q
N_P2 = int(os.environ.get("PF_NP2", "11"))   # value pairs of key block 0 whose exp stream runs in phase 2
N_B = int(os.environ.get("PF_NB", "18"))     # entries of that stream that run beside MFMAs 10..15
P1_DMA, P1_VR = (int(x) for x in os.environ.get("PF_P1MIX", "3,4").split(","))   # phase 1: a DMA / V read every n-th entry


def exp_pairs(buf, kb):
    """Per value pair of the S registers of key block kb: (exp a, exp b, add a, add b, cvt, mul a, mul b).  The
    accumulators hold q.k - m; the multiplies (exact mode) bring them to scale * log2(e) * (q.k - m)."""
    pairs = []
    for qb in range(2):
        for i in range(0, 16, 2):
            ra, rb = S(buf, kb, qb, i), S(buf, kb, qb, i + 1)
            ks = 2 * kb + (i >> 3)
            w = (i & 7) >> 1
            pairs.append((f"v_exp_f32 v{ra}, v{ra}", f"v_exp_f32 v{rb}, v{rb}",
                          f"v_add_f32 v{L[qb]}, v{L[qb]}, v{ra}", f"v_add_f32 v{L[qb] + 1}, v{L[qb] + 1}, v{rb}",
                          ("cvt", f"v_cvt_pk_bf16_f32 v{P(ks, qb, w)}, v{ra}, v{rb}"),
                          f"v_mul_f32 v{ra}, s{S_R}, v{ra}", f"v_mul_f32 v{rb}, s{S_R}, v{rb}"))
    return pairs


def flatten_pairs(pairs):
    """Skewed by one pair: the two exponentials of a pair are not neighbours and nothing reads an
    exponential in the instruction after it; in exact mode the multiplies of the NEXT pair sit behind the
    exponentials of this one."""
    out = []
    if pairs and EXACT:
        out += [x for x in (pairs[0][5], pairs[0][6]) if x]
    for n, (ea, eb, aa, ab, cv, ma, mb) in enumerate(pairs):
        out.append(ea)
        if EXACT and n + 1 < len(pairs):
            out.append(pairs[n + 1][5])
        if n > 0:
            out.append(pairs[n - 1][2])
        out.append(eb)
        if EXACT and n + 1 < len(pairs) and pairs[n + 1][6]:
            out.append(pairs[n + 1][6])
        if n > 0:
            out.append(pairs[n - 1][3])
            out.append(pairs[n - 1][4])
    if pairs:
        out += [pairs[-1][2], pairs[-1][3], pairs[-1][4]]
    return out


def exps_p2(buf):
    return flatten_pairs(exp_pairs(buf, 0)[:N_P2])


def exps_p1(buf):
    return flatten_pairs(exp_pairs(buf, 0)[N_P2:] + exp_pairs(buf, 1))


def plain(xs):
    return [x[1] if isinstance(x, tuple) else x for x in xs]


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
# placement knobs (experiments): MFMAs of phase 2 that carry the maxima; spacing of the V reads in phase 1
# (v_pk_fma_f32 / v_pk_add_f32 / v_dot2_f32_bf16 in the exp stream measured slower than the plain instructions
# they replace; see the experiments notes of the profiles directory.)
# timing-only early exits (nothing is stored): PF_EXIT=setup | fetched | prologue
EXIT_AT = os.environ.get("PF_EXIT", "")
# timing-only (wrong results): PF_SKIP=finish drops the output normalisation / transposition / stores of an
# item, PF_SKIP=epi the last tile's softmax and products (the next item's prefetch stays)
SKIP = set(filter(None, os.environ.get("PF_SKIP", "").split(",")))
# timing-only ablations (wrong results by construction): PF_ABLATE=dma,barrier,exp,max,vread,kread,mfma,vmwait
ABLATE = set(filter(None, os.environ.get("PF_ABLATE", "").split(",")))


def ablated(txt: str) -> bool:
    op = txt.split()[0]
    if "dma" in ABLATE and op.startswith("global_load_lds"):
        return True
    if "barrier" in ABLATE and op in ("s_barrier",):
        return True
    if "exp" in ABLATE and op in ("v_exp_f32", "v_fma_f32", "v_add_f32", "v_cvt_pk_bf16_f32"):
        return True
    if "max" in ABLATE and op in ("v_max3_f32",):
        return True
    if "vread" in ABLATE and op == "ds_read_b64_tr_b16":
        return True
    if "kread" in ABLATE and op == "ds_read_b128" and "a[19" in txt or "kread" in ABLATE and op == "ds_read_b128" and "a[2" in txt:
        return True
    if "mfma" in ABLATE and op.startswith("v_mfma"):
        return True
    if "vmwait" in ABLATE and txt.startswith("s_waitcnt vmcnt"):
        return True          # (data may be read before it has landed: timing only)
    if op.startswith("s_cbranch") and (".Lpf_slow" in txt or ".Lpf_resc" in txt):
        return True          # every timing build skips the data-dependent side paths
    return False


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
    kind, d, a, b, lab, c = m
    if kind in ("acc", "start_masked"):
        o.e(f"{MFMA} {d}, {a}, {b}, {d}")
        return
    if kind == "start_plain":
        o.e(f"{MFMA} {d}, {a}, {b}, {c}")
        return
    # chain start: C = the -m block (or 0), or C = the mask tile already sitting in the accumulator registers
    o.e(f"s_cmp_lg_u32 s{S_MASKED}, 0")
    o.e(f"s_cbranch_scc1 {lab}")
    o.e(f"{MFMA} {d}, {a}, {b}, {c}")
    o.label(f"{lab}_ret")
    _ool.append([f"{lab}:", f"  {MFMA} {d}, {a}, {b}, {d}", f"  s_branch {lab}_ret"])


# ---- building blocks -------------------------------------------------------------------------------
def mask_setup(o: Out, buf, tag, first=False, masked_phase=None):
    """A masked tile (j >= mask_from) presets S(buf) with -m (visible; 0 for the first tile of an item, whose
    chains start from 0) / -inf (hidden).  first: s_masked tells the chain starts which C to take and the
    block returns; otherwise the block continues into `masked_phase`, its own copy of phase 1 whose chains
    start from the preset tile (lines; it ends with a branch to the common phase 2)."""
    o.e(f"s_cmp_ge_u32 s{S_J}, s{S_MASKFROM}")
    if masked_phase is None:
        o.e(f"s_cselect_b32 s{S_MASKED}, 1, 0")
    o.e(f"s_cbranch_scc1 .Lpf_mask_{tag}")
    if masked_phase is None:
        o.label(f".Lpf_mask_{tag}_ret")
    blk = [f".Lpf_mask_{tag}:", f"  s_lshl_b32 s{S_T0}, s{S_J}, 6"]
    for qb in range(2):
        blk.append(f"  v_subrev_u32 v{T[qb]}, s{S_T0}, v{QPOS[qb]}")
    blk.append("  s_nop 1")
    for qb in range(2):
        vis = "0" if first else f"v{NMB(qb)}"
        for kb in range(2):
            for i0 in range(0, 16, 4):
                for k in range(4):
                    i = i0 + k
                    key = kb * 32 + (i & 3) + 8 * (i >> 2)
                    blk.append(f"  v_cmp_ge_i32 s[{S_CMP + 2 * k}:{S_CMP + 2 * k + 1}], v{T[qb]}, {key}")
                for k in range(4):
                    blk.append(f"  v_cndmask_b32 v{S(buf, kb, qb, i0 + k)}, v{NINF}, {vis}, "
                               f"s[{S_CMP + 2 * k}:{S_CMP + 2 * k + 1}]")
    blk.append("  s_nop 3")
    if masked_phase is None:
        blk.append(f"  s_branch .Lpf_mask_{tag}_ret")
    else:
        blk += masked_phase
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


def slow_max(o: Out, buf, tag):
    """Decision of the deferred rescale.  The tile maxima are relative to the running maxima already: any row
    above 8 / c sends the wave through the side path, which raises the maxima of all rows by max(tile maximum, 0):
    shifts the tile's accumulators and the -m blocks, records the factors and rescales the row sums."""
    o.e(f"v_max_f32 v{T[0]}, v{TM[0]}, v{TM[1]}")
    o.e(f"v_cmp_lt_f32 vcc, s{S_THR}, v{T[0]}" if EXACT else f"v_cmp_lt_f32 vcc, {EIGHT}, v{T[0]}")
    o.e(f"s_cbranch_vccnz .Lpf_slow_{tag}")
    o.label(f".Lpf_slow_{tag}_ret")
    blk = [f".Lpf_slow_{tag}:"]
    for qb in range(2):
        blk.append(f"  v_max_f32 v{T[qb]}, 0, v{TM[qb]}")
    if EXACT:
        for qb in range(2):
            blk.append(f"  v_mul_f32 v{AL[qb]}, s{S_R}, v{T[qb]}")      # the maxima are raw scores: c * shift
        for qb in range(2):
            blk.append(f"  v_exp_f32 v{AL[qb]}, -v{AL[qb]}")
    else:
        for qb in range(2):
            blk.append(f"  v_exp_f32 v{AL[qb]}, -v{T[qb]}")
    for qb in range(2):
        for kb in range(2):
            for i in range(16):
                r = S(buf, kb, qb, i)
                blk.append(f"  v_sub_f32 v{r}, v{r}, v{T[qb]}")
        for i in range(16):
            blk.append(f"  v_sub_f32 v{NMB(qb, i)}, v{NMB(qb, i)}, v{T[qb]}")
    for qb in range(2):
        blk += [f"  v_mul_f32 v{L[qb]}, v{L[qb]}, v{AL[qb]}",
                f"  v_mul_f32 v{L[qb] + 1}, v{L[qb] + 1}, v{AL[qb]}"]
    blk += [f"  s_mov_b32 s{S_RESC}, 1", "  s_nop 3", f"  s_branch .Lpf_slow_{tag}_ret"]
    _ool.append(blk)


def first_tile_max(o: Out):
    """First tile of an item (its chains started from 0): the running maxima become the tile maxima (not
    below -1e30, which keeps rows without any visible key free of inf - inf)."""
    for qb in range(2):
        o.e(f"v_max_f32 v{T[qb]}, {NEG_BIG}, v{TM[qb]}")
    for qb in range(2):
        for kb in range(2):
            for i in range(16):
                r = S(0, kb, qb, i)
                o.e(f"v_sub_f32 v{r}, v{r}, v{T[qb]}")
        o.e(f"v_sub_f32 v{NMB(qb, 0)}, 0, v{T[qb]}")
        for i in range(1, 16):
            o.e(f"v_mov_b32 v{NMB(qb, i)}, v{NMB(qb, 0)}")


def lds_ops_after(lines, marker):
    """Number of LDS instructions in `lines` behind the last line that contains `marker`."""
    hits = [i for i, ln in enumerate(lines) if marker in ln]
    if not hits:                     # timing-only builds may have dropped the reads
        return 0
    last = hits[-1]
    return sum(1 for ln in lines[last + 1:] if ln.strip().startswith("ds_"))


def rescale_check(o: Out, tag):
    o.e(f"s_cmp_lg_u32 s{S_RESC}, 0")
    o.e(f"s_cbranch_scc1 .Lpf_resc_{tag}")
    o.label(f".Lpf_resc_{tag}_ret")
    blk = [f".Lpf_resc_{tag}:", "  s_nop 15", "  s_nop 15"]
    tmp = [T[0], T[1]]     # v[252:255] carry the next tiles' table entries at this point
    for qb in range(2):
        for d in range(4):
            for i0 in range(0, 16, 2):
                for k in range(2):
                    blk.append(f"  v_accvgpr_read_b32 v{tmp[k]}, a{O(qb, d, i0 + k)}")
                blk.append("  s_nop 1")
                for k in range(2):
                    blk.append(f"  v_mul_f32 v{tmp[k]}, v{tmp[k]}, v{AL[qb]}")
                blk.append("  s_nop 0")
                for k in range(2):
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
def q_fetch(o: Out, rw0_sreg, areg_base):
    """Global loads of the 16 Q fragments of the item whose first row is in rw0_sreg straight into 64
    accumulation registers from areg_base (fragment (qb, s) at + 32 qb + 4 s): source row
    min(rw0 + 32 qb + r, n_q - 1), bytes 32 s + 16 h of the head row.  T[0], T[2], T[3] are used;
    T[4] = lane & 31 and T[5] = 16 (lane >> 5) must be set (lane_consts)."""
    for qb in range(2):
        o.e(f"v_add_u32 v{T[0]}, s{rw0_sreg}, v{T[4]}")
        if qb:
            o.e(f"v_add_u32 v{T[0]}, 32, v{T[0]}")
        o.e(f"v_min_u32 v{T[0]}, s{S_NQ1}, v{T[0]}")
        o.e(f"v_mad_u32_u24 v{T[2 + qb]}, v{T[0]}, s{S_QSTR}, v{T[5]}")
    for qb in range(2):
        for sk in range(8):
            r = areg_base + 32 * qb + 4 * sk
            o.e(f"global_load_dwordx4 a[{r}:{r + 3}], v{T[2 + qb]}, s[{S_QSEQ}:{S_QSEQ + 1}] offset:{32 * sk}")


def tile_fetch_const(o: Out, tile_const, kind, slot, offregs, tag, guard_nt):
    """Straight-line fetch of tile `tile_const` (prologue of an item / prefetch for the next item)."""
    if guard_nt:
        o.e(f"s_cmp_le_u32 s{S_NT}, {tile_const}")
        o.e(f"s_cbranch_scc1 .Lpf_skip_{tag}")
    o.e(f"s_mov_b32 s{S_T0}, {tile_const}")
    tail_fix(o, S_T0, offregs, tag)
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
        o.label(f".Lpf_skip_{tag}")


def read_descriptor(o: Out, item_sreg, into_next=False):
    """Item descriptor (8 dwords at s_desc + 32 item): nt, mask_from, rw0 + shift, rw0, rows."""
    o.e(f"s_lshl_b32 s{S_T0}, s{item_sreg}, 5")
    o.e(f"s_add_u32 s{S_T0}, s{S_T0}, s{S_DESC}")
    # the address register must not be one of the destinations: the second read is issued while the first
    # may already have returned (rarely, when the LDS queue holds the issue back), and would then take a
    # descriptor word for its address
    o.e(f"v_mov_b32 v{T[5]}, s{S_T0}")
    o.e(f"ds_read_b128 v[250:253], v{T[5]}")
    o.e(f"ds_read_b32 v254, v{T[5]} offset:16")
    o.e("s_waitcnt lgkmcnt(0)")
    o.e("s_nop 1")
    if into_next:
        o.e(f"v_readfirstlane_b32 s{S_T0 + 1}, v253")            # rw0 of the next item
    else:
        o.e(f"v_readfirstlane_b32 s{S_NT}, v250")
        o.e(f"v_readfirstlane_b32 s{S_MASKFROM}, v251")
        o.e(f"v_readfirstlane_b32 s{S_ITEMBASE}, v252")
        o.e(f"v_readfirstlane_b32 s{S_RW0}, v253")
        o.e(f"v_readfirstlane_b32 s{S_ROWS}, v254")


def lane_consts(o: Out):
    """T[4] = lane & 31, T[5] = 16 * (lane >> 5) (temporaries: recomputed where needed)."""
    o.e(f"v_mbcnt_lo_u32_b32 v{T[0]}, -1, 0")
    o.e(f"v_mbcnt_hi_u32_b32 v{T[0]}, -1, v{T[0]}")
    o.e(f"v_and_b32 v{T[4]}, 31, v{T[0]}")
    o.e(f"v_lshrrev_b32 v{T[5]}, 5, v{T[0]}")
    o.e(f"v_lshlrev_b32 v{T[5]}, 4, v{T[5]}")


def gen_setup(o: Out):
    o.c("---- once per workgroup: operands to fixed registers, lane constants ----")
    ops = [(S_KBASE, 64), (S_VBASE, 64), (S_C, 32), (S_TAIL, 32), (S_PIECE, 32), (S_STG, 32), (S_TBL, 32),
           (S_DESC, 32), (S_NITEMS, 32), (S_QSEQ, 64), (S_OSEQ, 64), (S_QSTR, 32), (S_OSTR, 32), (S_NQ1, 32), (S_MODE, 32)]
    for i, (r, bits) in enumerate(ops):
        if bits == 64:
            o.e(f"s_mov_b64 s[{r}:{r + 1}], %{i}")
        else:
            o.e(f"s_mov_b32 s{r}, %{i}")
    o.e(f"s_mov_b32 s{S_ITEM}, 0")
    o.e(f"s_mov_b32 s{S_RESC}, 0")
    o.e(f"v_mbcnt_lo_u32_b32 v{T[0]}, -1, 0")
    o.e(f"v_mbcnt_hi_u32_b32 v{T[0]}, -1, v{T[0]}")
    o.e(f"v_lshlrev_b32 v{PADDR}, 2, v{T[0]}")
    o.e(f"v_add_u32 v{PADDR}, s{S_STG}, v{PADDR}")
    for pi, r in enumerate([KB[0], KB[1], VB[0], VB[1]]):
        o.e(f"ds_read_b32 v{r}, v{PADDR} offset:{PARAM_OFF + pi * 256}")
    o.e(f"v_mov_b32 v{NINF}, 0xff800000")
    o.c("first item: its descriptor; its Q fragments travel to the K fragment registers (scaled into place later)")
    read_descriptor(o, S_ITEM)
    o.c("words 5, 6 of the first descriptor: c and 8 / c (the operand list is as long as the asm statement allows)")
    o.e(f"ds_read_b32 v{T[0]}, v{T[5]} offset:20")
    o.e(f"ds_read_b32 v{T[1]}, v{T[5]} offset:24")
    o.e("s_waitcnt lgkmcnt(0)")
    o.e("s_nop 1")
    o.e(f"v_readfirstlane_b32 s{S_R}, v{T[0]}")
    o.e(f"v_readfirstlane_b32 s{S_THR}, v{T[1]}")
    o.c("first item: its Q fragments were requested as one tile image (the K tile layout) into the wave's staging")
    o.c("area by the code in front of this body, AHEAD of the first K/V tiles: they are read back while those fly")
    o.e(f"ds_read_b32 v{T[2]}, v{PADDR} offset:{PARAM_OFF + 14 * 256}")
    o.e(f"ds_read_b32 v{T[3]}, v{PADDR} offset:{PARAM_OFF + 15 * 256}")
    o.e("s_waitcnt lgkmcnt(0)")
    o.e(f"s_waitcnt vmcnt({FIRST_KV_PIECES})")
    for qb in range(2):
        for sk in range(8):
            o.e(f"ds_read_b128 {rng('a', K(qb, sk), 4)}, v{T[2 + (sk & 1)]} offset:{8192 * qb + 512 * (sk >> 1)}")
    if EXIT_AT == "setup":
        o.e("s_waitcnt vmcnt(0)")
        o.e("s_branch .Lpf_end")


def q_prescale(o: Out):
    """Q fragments (bf16 pairs) from the K fragment registers, where the item's fetch put them while the item
    before still used the Q registers, to their own registers: a plain move for the factor 1 (any positive
    softmax scale in exact mode), else times the factor in s_c on a side path."""
    tmp = [T[0], T[1], T[2], T[3]]
    scaled = []
    for i0 in range(0, 64, 2):
        for k in range(2):
            scaled.append(f"v_accvgpr_read_b32 v{tmp[2 * k]}, a{K(0, 0) + i0 + k}")
        scaled.append("s_nop 0")
        for k in range(2):
            scaled.append(f"v_lshlrev_b32 v{tmp[2 * k + 1]}, 16, v{tmp[2 * k]}")
        for k in range(2):
            scaled.append(f"v_and_b32 v{tmp[2 * k]}, 0xffff0000, v{tmp[2 * k]}")
        for k in range(2):
            scaled.append(f"v_mul_f32 v{tmp[2 * k + 1]}, s{S_C}, v{tmp[2 * k + 1]}")
            scaled.append(f"v_mul_f32 v{tmp[2 * k]}, s{S_C}, v{tmp[2 * k]}")
        for k in range(2):
            scaled.append(f"v_cvt_pk_bf16_f32 v{tmp[2 * k]}, v{tmp[2 * k + 1]}, v{tmp[2 * k]}")
        scaled.append("s_nop 0")
        for k in range(2):
            scaled.append(f"v_accvgpr_write_b32 a{Q(0, 0) + i0 + k}, v{tmp[2 * k]}")
    if not EXACT:
        for ln in scaled:
            o.e(ln)
        return
    o.e(f"s_cmp_lg_u32 s{S_C}, 0x3f800000")
    o.e("s_cbranch_scc1 .Lpf_qscale")
    for i in range(64):
        o.e(f"v_accvgpr_mov_b32 a{Q(0, 0) + i}, a{K(0, 0) + i}")
    o.label(".Lpf_qscale_ret")
    _ool.append([".Lpf_qscale:"] + ["  " + ln for ln in scaled] + ["  s_branch .Lpf_qscale_ret"])


def gen_item_prologue(o: Out):
    """Start of an item.  On entry: descriptor scalars loaded, Q fragments of the item in flight to (or landed
    in) the K fragment registers."""
    o.label(".Lpf_item")
    for i in range(4):
        o.e(f"ds_read_b32 v{DOFFK[i]}, v{PADDR} offset:{PARAM_OFF + (4 + i) * 256}")
        o.e(f"ds_read_b32 v{DOFFV[i]}, v{PADDR} offset:{PARAM_OFF + (4 + i) * 256}")
    o.e(f"ds_read_b32 v{QPOS[0]}, v{PADDR} offset:{PARAM_OFF + 12 * 256}")
    o.e(f"ds_read_b32 v{QPOS[1]}, v{PADDR} offset:{PARAM_OFF + 13 * 256}")
    o.e(f"s_mov_b32 s{S_J}, 0")
    for qb in range(2):
        o.e(f"v_mov_b32 v{L[qb]}, 0")
        o.e(f"v_mov_b32 v{L[qb] + 1}, 0")
    for i in range(128):
        o.e(f"v_accvgpr_write_b32 a{i}, 0")
    o.e("s_waitcnt lgkmcnt(0)")
    for qb in range(2):
        o.e(f"v_add_u32 v{QPOS[qb]}, s{S_ITEMBASE}, v{QPOS[qb]}")
    # K(0), V(0) -> slot 0 and K(1) -> slot 1 are in flight or have landed: the code in front of the asm body
    # requested them for the first item, the epilogue of the item before for every later one
    o.e("s_waitcnt vmcnt(0)")
    q_prescale(o)
    o.e("s_barrier")
    if EXIT_AT == "fetched":
        o.e("s_branch .Lpf_end")
    o.e(f"s_cmp_eq_u32 s{S_NT}, 0")
    o.e("s_cbranch_scc1 .Lpf_fin")
    for ln in k_reads(0):
        o.e(ln)
    o.e("s_waitcnt lgkmcnt(0)")
    o.e("s_barrier")
    tile_fetch_const(o, 2, "K", 0, DOFFK, "pro_k2", False)
    mask_setup(o, 0, "pro", first=True)
    for m in qk_mfmas(0, ".Lpf_mm_pro", cinit=False):
        emit_mfma(o, m)
    o.e("s_nop 15")
    o.e("s_nop 7")
    o.c("tile 0: its maxima become the running maxima; accumulators and -m blocks are shifted to them")
    for ln in max_stream(0):
        o.e(ln)
    first_tile_max(o)
    for ln in k_reads(1):
        o.e(ln)
    for x in plain(exps_p2(0)):
        o.e(x)
    lk: list[str] = []
    table_lookup(lk, 24, 8)          # next iteration j = 1 fetches K(3) and V(1); s_j is still 0 here
    for ln in lk:
        o.e(ln)
    o.e("s_waitcnt lgkmcnt(0)")
    o.e("s_waitcnt vmcnt(0)")
    o.e("s_barrier")
    o.e(f"s_mov_b32 s{S_J}, 1")
    if EXIT_AT == "prologue":
        o.e("s_branch .Lpf_end")
    o.e(f"s_cmp_le_u32 s{S_NT}, 1")
    o.e("s_cbranch_scc1 .Lpf_epi0")


def gen_body(o: Out, p):
    """Iteration j with j & 1 == p: S(j) -> buffer p; the pending tile j-1 sits in buffer 1-p."""
    q = 1 - p
    tag = f"b{p}"
    o.label(f".Lpf_body{p}")
    # ---- phase 1 ----
    vr = v_reads(q, [0, 1])
    ex = plain(exps_p1(q))
    # V(j) and K(j+2) go to slot p.  The tile that holds the end of the sequence switches the wave's
    # source offsets to the clamped set (rare, checked here: branches stay out of the MFMA stream).
    # K(j+2) may lie behind the last tile: its table entry then repeats the last tile (a harmless fetch
    # into a slot nobody reads again).
    tail_fix(o, S_J, DOFFV, f"{tag}_v")
    o.e(f"s_add_u32 s{S_T0}, s{S_J}, 2")
    tail_fix(o, S_T0, DOFFK, f"{tag}_k")
    dma = [f"v_readfirstlane_b32 s{S_T0 + 6}, v254", f"v_readfirstlane_b32 s{S_T0 + 7}, v255",
           f"v_readfirstlane_b32 s{S_T0 + 2}, v252", f"v_readfirstlane_b32 s{S_T0 + 3}, v253",
           # one filler (two instructions): nothing that writes SCC may come between the add and the
           # add-with-carry
           f"s_add_u32 s54, s{S_VBASE}, s{S_T0 + 6}\n  s_addc_u32 s55, s{S_VBASE + 1}, s{S_T0 + 7}"]
    dma += dma_pieces("V", p, DOFFV)
    dma += [f"s_add_u32 s54, s{S_KBASE}, s{S_T0 + 2}\n  s_addc_u32 s55, s{S_KBASE + 1}, s{S_T0 + 3}"]
    dma += dma_pieces("K", p, DOFFK)
    fill1: list = []
    vi = di = 0
    for k, x in enumerate(ex):
        fill1.append(x)
        if k % P1_DMA == 1 and di < len(dma):
            fill1.append(dma[di])
            di += 1
        if k % P1_VR == 2 % P1_VR and vi < len(vr):
            fill1.append(vr[vi])
            vi += 1
    fill1 += dma[di:] + vr[vi:]
    # two copies of phase 1: the common one starts its S chains from the -m blocks; the one for masked tiles
    # (out of line, behind the block that presets S) starts them from S itself.  One branch per tile.
    om = Out()
    om.hot = o.hot
    interleave(om, qk_mfmas(p, f".Lpf_mm_{tag}", mode="masked"), fill1, tag + "p1m")
    om.e(f"s_branch .Lpf_p2_{tag}")
    mask_setup(o, p, tag, masked_phase=om.lines)
    interleave(o, qk_mfmas(p, f".Lpf_mm_{tag}", mode="plain"), fill1, tag + "p1")
    o.label(f".Lpf_p2_{tag}")
    o.e("s_waitcnt lgkmcnt(0)")
    # ---- phase 2: O += V(j-1)^T P(j-1) in four k-steps of 8 MFMAs ----
    pv = pv_mfmas()
    mx = max_stream(p)
    kr = k_reads(q)
    ex2 = exps_p2(p)
    # (a) MFMAs 0..9: the addresses of the next tiles (v[252:255] are free once phase 1 has taken this tile's),
    #     the row maxima of S(j) (the last S accumulator was written by the final MFMA of phase 1 and is read
    #     more than 16 instructions later), half of the K reads.  LDS reads go first in every part: the wait
    #     at the end of the tile should find them done.
    lk: list[str] = []
    table_lookup(lk, 24, 8)          # iteration j+1 fetches K(j+3) and V(j+1); s_j is still j here
    fa: list = list(lk)
    ki = 0
    for k, x in enumerate(mx):
        fa.append(x)
        if k % 4 == 3 and ki < 8:
            fa.append(kr[ki])
            ki += 1
    interleave(o, pv[0:10], fa, tag + "p2a")
    slow_max(o, p, tag)
    # (b) MFMAs 10..15: k-step 2 of V(j-1) into the registers of k-step 0 (its MFMAs 0..7 have issued), the
    #     start of the exp stream; the P fragments of k-steps 0, 1 are still being read: packing waits
    n_b = N_B
    eb = [x for x in ex2[:n_b] if not isinstance(x, tuple)]
    late = [x[1] for x in ex2[:n_b] if isinstance(x, tuple)]
    start = len(o.lines)
    vr2 = v_reads(q, [2])
    interleave(o, pv[10:16], vr2 + eb, tag + "p2b")
    o.e(f"s_waitcnt lgkmcnt({lds_ops_after(o.lines[start:], vr2[-1])})")
    # (c) MFMAs 16..23: k-step 3 into the registers of k-step 1, K reads, packing, more of the exp stream
    rest = plain(ex2[n_b:])
    n_c = (len(rest) + len(late)) * 8 // 16
    fc = late + rest[:max(0, n_c - len(late))]
    fd = rest[max(0, n_c - len(late)):]
    start = len(o.lines)
    vr3 = v_reads(q, [3])
    interleave(o, pv[16:24], vr3 + kr[ki:ki + 4] + fc, tag + "p2c")
    ki += 4
    o.e(f"s_waitcnt lgkmcnt({lds_ops_after(o.lines[start:], vr3[-1])})")
    # (d) MFMAs 24..31: the last K reads, the rest of the exp stream
    interleave(o, pv[24:32], kr[ki:] + fd, tag + "p2d")
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
    """After the last tile of an item (it sits in buffer qbuf): finish its softmax and run its products; once
    every wave has read the tile's last V k-step the ring is free, and tiles 0 / 1 of the NEXT item are fetched
    while the last products and the output stores run."""
    o.label(f".Lpf_epi{qbuf}")
    for ln in v_reads(qbuf, [0, 1]):
        o.e(ln)
    for x in plain(exps_p1(qbuf)):
        if "epi" not in SKIP:
            o.e(x)
    o.e("s_waitcnt lgkmcnt(0)")
    o.e("s_nop 1")
    pv = pv_mfmas()
    if "epi" in SKIP:
        pv = ["s_nop 0"] * 32
    for m in pv[0:9]:
        o.e(m)
    for ln in v_reads(qbuf, [2]):
        o.e(ln)
    for m in pv[9:16]:
        o.e(m)
    o.e("s_waitcnt lgkmcnt(0)")
    o.e(pv[16])
    for ln in v_reads(qbuf, [3]):
        o.e(ln)
    for m in pv[17:24]:
        o.e(m)
    o.e("s_waitcnt lgkmcnt(0)")
    o.e("s_barrier")
    o.e(f"s_add_u32 s{S_LAST}, s{S_ITEM}, 1")
    o.e(f"s_cmp_ge_u32 s{S_LAST}, s{S_NITEMS}")
    o.e(f"s_cbranch_scc1 .Lpf_epi{qbuf}_nonext")
    o.c("source offsets back to the unclamped set, then K0, V0 -> slot 0, K1 -> slot 1 of the next item")
    for i in range(4):
        o.e(f"ds_read_b32 v{DOFFK[i]}, v{PADDR} offset:{PARAM_OFF + (4 + i) * 256}")
        o.e(f"ds_read_b32 v{DOFFV[i]}, v{PADDR} offset:{PARAM_OFF + (4 + i) * 256}")
    o.e("s_waitcnt lgkmcnt(0)")
    tile_fetch_const(o, 0, "K", 0, DOFFK, f"epi{qbuf}_k0", False)
    tile_fetch_const(o, 0, "V", 0, DOFFV, f"epi{qbuf}_v0", False)
    tile_fetch_const(o, 1, "K", 1, DOFFK, f"epi{qbuf}_k1", False)
    o.c("Q fragments of the next item -> the K fragment registers (dead until the next item's first tile)")
    read_descriptor(o, S_LAST, into_next=True)
    lane_consts(o)
    q_fetch(o, S_T0 + 1, K(0, 0))
    o.label(f".Lpf_epi{qbuf}_nonext")
    o.e("s_nop 3")
    for m in pv[24:32]:
        o.e(m)
    o.e("s_branch .Lpf_fin")


def partial_state_store():
    """Key-range split: the wave's rows leave as an unnormalised state in the order the registers hold it, one
    17,408-byte unit per (wave, key range) at s[S_OSEQ:S_OSEQ+1]:
        [qb 2][d 4][half 2][lane 64][8 bf16]   lane (r, h) holds accumulators i = 8 half .. 8 half + 7 of block
                                               (qb, d): dims 32 d + 8 (i >> 2) + 4 h + (i & 3) of row 32 qb + r
        [qb 2][lane 64][m, l]  fp32            at byte 16384 (the same pair in both lanes of a row)
    so every store instruction writes 1 KiB (512 B) of consecutive memory; attn_prefill_merge_image_kernel reads
    the units back.  The accumulators are rounded to bf16 (a relative 2^-9 per key range, the size of the output's
    own rounding; bf16 has the exponent range of fp32, so unnormalised sums fit); m and l stay fp32.  The state
    traffic bounds a split launch (every byte is written and read once, at memory speed), which is why it is
    halved here.  All 64 rows are written (rows behind the end of the sequence repeat its last row; the merge
    does not store them).  Out of line; v[0:67] are free here."""
    b = [".Lpf_fin_part:"]
    b += [f"  s_mov_b64 s[{S_T0 + 2}:{S_T0 + 3}], s[{S_OSEQ}:{S_OSEQ + 1}]",
          f"  v_mbcnt_lo_u32_b32 v{T[0]}, -1, 0", f"  v_mbcnt_hi_u32_b32 v{T[0]}, -1, v{T[0]}",
          f"  v_lshlrev_b32 v{T[3]}, 4, v{T[0]}",                                       # 16 bytes per lane
          f"  v_lshlrev_b32 v{T[2]}, 3, v{T[0]}", f"  v_add_u32 v{T[2]}, 0x4000, v{T[2]}"]  # (m, l) pairs behind the image
    k = 0
    for qb in range(2):
        b += [f"  v_sub_f32 v{64 + 2 * qb}, 0, v{NMB(qb)}", f"  v_mov_b32 v{65 + 2 * qb}, v{L[qb]}"]
        if EXACT:
            b += [f"  v_mul_f32 v{64 + 2 * qb}, s{S_R}, v{64 + 2 * qb}"]     # the running maximum in log2 units
        for d in range(4):
            pk = 16 + 8 * (k % 6)          # packed registers rotate: a store reads its data some time after issue
            k += 1
            for i in range(16):
                b.append(f"  v_accvgpr_read_b32 v{i}, a{O(qb, d, i)}")
            b.append("  s_nop 1")
            for i in range(8):
                b.append(f"  v_cvt_pk_bf16_f32 v{pk + i}, v{2 * i}, v{2 * i + 1}")
            b.append("  s_nop 0")
            for half in range(2):
                b.append(f"  global_store_dwordx4 v{T[3]}, v[{pk + 4 * half}:{pk + 4 * half + 3}], "
                         f"s[{S_T0 + 2}:{S_T0 + 3}] offset:{(d & 1) * 2048 + half * 1024}")
            if d & 1:
                b.append(f"  v_add_u32 v{T[3]}, 0x1000, v{T[3]}")
        b.append(f"  global_store_dwordx2 v{T[2]}, v[{64 + 2 * qb}:{65 + 2 * qb}], s[{S_T0 + 2}:{S_T0 + 3}] offset:{qb * 512}")
    b.append("  s_branch .Lpf_fin_next")
    _ool.append(b)


def gen_finish(o: Out):
    o.label(".Lpf_fin")
    if "finish" in SKIP:
        o.e("s_branch .Lpf_fin_next")
    o.c("denominators: add the share of the lane that holds the other half of each row; 1 / l (0 for rows without keys)")
    o.e("s_nop 15")
    o.e("s_nop 15")
    for qb in range(2):
        o.e(f"v_add_f32 v{L[qb]}, v{L[qb]}, v{L[qb] + 1}")
    for qb in range(2):
        o.e(f"v_mov_b32 v{T[qb]}, v{L[qb]}")
    o.e("s_nop 1")
    for qb in range(2):
        o.e(f"v_permlane32_swap_b32 v{L[qb]}, v{T[qb]}")
    for qb in range(2):
        o.e(f"v_add_f32 v{L[qb]}, v{L[qb]}, v{T[qb]}")
    o.e(f"s_cmp_lg_u32 s{S_MODE}, 0")
    o.e("s_cbranch_scc1 .Lpf_fin_part")
    partial_state_store()
    for qb in range(2):
        o.e(f"v_rcp_f32 v{AL[qb]}, v{L[qb]}")
        o.e(f"v_cmp_lt_f32 vcc, 0, v{L[qb]}")
        o.e("s_nop 1")
        o.e(f"v_cndmask_b32 v{AL[qb]}, 0, v{AL[qb]}, vcc")
    o.c("O^T -> bf16 rows in the wave's staging area (row qb*32 + (lane & 31), 272-byte row stride), then")
    o.c("whole 256-byte row segments to memory, 4 rows per store instruction; rows behind the sequence end are masked")
    lane_consts(o)                                                  # T[0] lane, T[4] = r, T[5] = 16 h
    o.e(f"v_lshrrev_b32 v{T[5]}, 1, v{T[5]}")                       # 8 h
    o.e(f"s_movk_i32 s{S_T0 + 4}, 272")
    o.e(f"v_mad_u32_u24 v{QPOS[0]}, v{T[4]}, s{S_T0 + 4}, v{T[5]}")
    o.e(f"v_add_u32 v{QPOS[0]}, s{S_STG}, v{QPOS[0]}")              # staging write address of the lane
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
                o.e(f"ds_write_b64 v{QPOS[0]}, v[{T[4]}:{T[5]}] offset:{qb * 32 * 272 + d * 64 + g * 16}")
    o.e(f"s_mul_i32 s{S_T0}, s{S_RW0}, s{S_OSTR}")
    o.e(f"s_mul_hi_u32 s{S_T0 + 1}, s{S_RW0}, s{S_OSTR}")
    o.e(f"s_add_u32 s{S_T0 + 2}, s{S_OSEQ}, s{S_T0}")
    o.e(f"s_addc_u32 s{S_T0 + 3}, s{S_OSEQ + 1}, s{S_T0 + 1}")
    o.e(f"v_mbcnt_lo_u32_b32 v{T[0]}, -1, 0")
    o.e(f"v_mbcnt_hi_u32_b32 v{T[0]}, -1, v{T[0]}")
    o.e(f"v_lshrrev_b32 v{T[1]}, 4, v{T[0]}")                       # row inside a group of 4
    o.e(f"v_and_b32 v{T[2]}, 15, v{T[0]}")
    o.e(f"v_lshlrev_b32 v{T[2]}, 4, v{T[2]}")                       # 16-byte piece of the row
    o.e(f"v_mad_u32_u24 v{T[3]}, v{T[1]}, s{S_T0 + 4}, v{T[2]}")
    o.e(f"v_add_u32 v{T[3]}, s{S_STG}, v{T[3]}")                    # staging read address
    o.e(f"v_mad_u32_u24 v{T[4]}, v{T[1]}, s{S_OSTR}, v{T[2]}")      # global offset of the lane's piece, group 0
    o.e(f"s_lshl_b32 s{S_T0 + 5}, s{S_OSTR}, 2")                    # 4 rows further
    o.e("s_waitcnt lgkmcnt(0)")
    for it in range(16):
        o.e(f"ds_read_b128 v[{4 * it}:{4 * it + 3}], v{T[3]} offset:{it * 4 * 272}")
    for it in range(16):
        o.e(f"s_sub_i32 s{S_T0}, s{S_ROWS}, {4 * it}")
        o.e(f"v_cmp_gt_i32 vcc, s{S_T0}, v{T[1]}")
        o.e(f"s_waitcnt lgkmcnt({15 - it})")
        o.e(f"s_and_saveexec_b64 s[{S_SAVE}:{S_SAVE + 1}], vcc")
        o.e(f"global_store_dwordx4 v{T[4]}, v[{4 * it}:{4 * it + 3}], s[{S_T0 + 2}:{S_T0 + 3}]")
        o.e(f"s_mov_b64 exec, s[{S_SAVE}:{S_SAVE + 1}]")
        o.e(f"v_add_u32 v{T[4]}, s{S_T0 + 5}, v{T[4]}")
    o.label(".Lpf_fin_next")
    o.c("next item")
    o.e(f"s_add_u32 s{S_ITEM}, s{S_ITEM}, 1")
    o.e(f"s_cmp_ge_u32 s{S_ITEM}, s{S_NITEMS}")
    o.e("s_cbranch_scc1 .Lpf_done")
    read_descriptor(o, S_ITEM)
    o.e("s_branch .Lpf_item")
    o.label(".Lpf_done")


def generate() -> str:
    o = Out()
    gen_setup(o)
    gen_item_prologue(o)
    o.hot = True
    gen_body(o, 1)
    gen_body(o, 0)
    o.hot = False
    gen_epilogue(o, 0)
    gen_epilogue(o, 1)
    gen_finish(o)
    o.e("s_branch .Lpf_end")
    for blk in _ool:
        o.lines += blk
    o.label(".Lpf_end")
    text = "\n".join(o.lines) + "\n"
    check_scc(text)
    return text


def check_scc(text: str):
    """Static checks of the generated code (tools/check_prefill_asm.py): SCC consumers directly behind their
    producers, no use of a register with a load outstanding into it, no early use of MFMA results.  The
    interleaver places instructions between others, so these are checked on its output."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import check_prefill_asm
    errs = check_prefill_asm.check(text)
    if errs:
        raise AssertionError("generated loop fails the static checks:\n  " + "\n  ".join(errs[:20]))


def clobbers() -> str:
    regs = [f"v{i}" for i in range(256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(36, 86)]
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
