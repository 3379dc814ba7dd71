#!/usr/bin/env python3
"""Static checks of the generated prefill loop (text of one asm body, one instruction per line).

The generator places instructions between others, so dependences that the hardware does not interlock are
checked on its OUTPUT:

  1. SCC: every consumer (s_addc_u32, s_cselect_b32, s_cbranch_scc*) directly follows its producer.
  2. Outstanding loads: between an LDS / global load into vector registers and the s_waitcnt that covers it, no
     instruction may read or write those registers -- including as an address (a read issued behind an LDS
     read into its own address register takes the loaded word for an address once its issue is held back).
  3. MFMA results: an instruction other than an MFMA that accumulates into the same registers may touch the
     destination registers of an MFMA only MFMA_WAIT instruction slots later.
  4. Short fixed distances: the result of a transcendental instruction is not read by the next instruction; a
     register written by a vector instruction is read by v_permlane32_swap at the earliest two slots later and
     by v_readfirstlane one slot later; a scalar register written by a vector instruction (v_readfirstlane,
     v_cmp) is used by a memory instruction as address base at the earliest five slots later; an MFMA reads an
     operand register written by a vector (non-MFMA) instruction at the earliest two slots later.
  5. Packed fp32 with half selection: a v_pk_*_f32 instruction whose op_sel takes the UPPER half of a register pair
     for its LOWER result must not be followed directly by an MFMA.  Measured on gfx950
     (tools/micro/pk_fma_opsel.hip): with an MFMA in the very next slot one sixteenth of the (lane, result) pairs
     come out as if the selected operand had not been read, whatever wrote that operand and however long before;
     one wait state or one other instruction in between, an MFMA in front instead of behind, or the form without
     half selection (the pair holds the value twice) are all clean.  A loop scheduler that places such an
     instruction as the last filler of an MFMA gap produces this "now and then".

The walk is linear over the text.  A label that is only reached by branches starts from the union of the states
at the branches to it; a label reached by falling through joins that union with the fall-through state.
    python tools/check_prefill_asm.py [file.inc]      (default: the generated csrc/attn_prefill_loop.inc)
"""
from __future__ import annotations

import os
import re
import sys

MFMA_WAIT = 12          # 8-pass XDL op: destination readable by other instructions after 11 wait states
_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
_SCC_WRITERS = ("s_cmp", "s_add", "s_sub", "s_and", "s_or", "s_xor", "s_lshl", "s_lshr", "s_ashr", "s_bfe", "s_min",
                "s_max", "s_abs", "s_not")


def regs_of(operand: str) -> set:
    out = set()
    for m in _REG.finditer(operand):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
    return out


def split_operands(rest: str) -> list:
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def parse(line: str):
    """-> (op, dst regs, src regs, kind) with kind in {'lds_load', 'vm_load', 'mfma', 'other'}."""
    parts = line.split(None, 1)
    op = parts[0]
    ops = split_operands(parts[1]) if len(parts) > 1 else []
    ops = [o for o in ops if not o.startswith(("offset:", "op_sel", "neg_", "sc1", "nt", "off"))
           or _REG.search(o.split()[0] if o else "")]
    ops = [o.split(" offset:")[0].split(" sc1")[0] for o in ops]
    allr = [regs_of(o) for o in ops]
    if op.startswith("v_mfma"):
        return op, allr[0], set().union(*allr[1:]), "mfma"
    if op.startswith("ds_read"):
        return op, allr[0], set().union(*allr[1:]), "lds_load"
    if op.startswith("ds_write"):
        return op, set(), set().union(*allr), "lds_other"
    if op.startswith("global_load_lds"):
        return op, set(), set().union(*allr), "vm_other"
    if op.startswith("global_load"):
        return op, allr[0], set().union(*allr[1:]), "vm_load"
    if op.startswith("global_store"):
        return op, set(), set().union(*allr), "vm_other"
    if op.startswith("v_permlane"):
        both = set().union(*allr)
        return op, both, both, "other"
    if op.startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
        return op, set(), set().union(*allr), "other"
    if op.startswith("v_"):
        return op, allr[0] if allr else set(), set().union(*allr[1:]) if len(allr) > 1 else set(), "other"
    return op, set(), set(), "scalar"


_SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
_TRANS = ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32")


def sregs_of(text: str) -> set:
    out = set()
    for m in _SREG.finditer(text):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_short_distances(lines: list) -> list:
    """Check 4 (linear; labels and branches reset the history: the code behind them does not depend on
    results produced right in front of them)."""
    errors = []
    hist: list = []          # (slots ago is implied by position) entries: (op, vector dst, scalar dst)
    for i, ln in enumerate(lines):
        if ln.endswith(":") or ln.split()[0] in ("s_branch",) or ln.split()[0].startswith("s_cbranch"):
            hist = []
            continue
        op, dst, src, kind = parse(ln)
        if op == "s_nop":
            hist += [("s_nop", set(), set(), False)] * (int(ln.split()[1]) + 1)
            continue
        ops = split_operands(ln.split(None, 1)[1]) if " " in ln else []
        sdst = sregs_of(ops[0]) if ops and op.startswith(("v_readfirstlane", "v_cmp")) else set()

        def written_within(n, regs, vector=True):
            for back, (hop, hv, hs, _) in enumerate(reversed(hist[-n:]) if n else []):
                if hop.startswith("v_") and ((hv if vector else hs) & regs):
                    return hop
            return None
        if hist and hist[-1][0].startswith(_TRANS) and (hist[-1][1] & src):
            errors.append(f"line {i}: '{ln}' reads the result of the transcendental instruction right in front of it")
        if op.startswith("v_permlane") and written_within(2, src):
            errors.append(f"line {i}: '{ln}' reads a register a vector instruction wrote less than two slots before")
        if kind == "mfma":
            w = written_within(2, src | dst)
            if w and not w.startswith("v_mfma"):
                errors.append(f"line {i}: '{ln}' reads an operand a vector instruction ({w}) wrote less than two slots before")
            if hist and hist[-1][0].startswith("v_pk_") and hist[-1][0].endswith("_f32") and hist[-1][3]:
                errors.append(f"line {i}: '{ln}' directly follows a packed fp32 instruction that selects an upper "
                              f"half for its lower result (op_sel): put one instruction or wait state between them")
        if op.startswith("v_readfirstlane") and written_within(1, src):
            errors.append(f"line {i}: '{ln}' reads a register the instruction in front of it wrote")
        if kind in ("vm_load", "vm_other"):
            base = sregs_of(" ".join(ops))
            if written_within(5, base, vector=False):
                errors.append(f"line {i}: '{ln}' uses a scalar register a vector instruction wrote less than five slots before")
        m_sel = re.search(r"op_sel:\[([01, ]+)\]", ln)
        half_sel = bool(m_sel and "1" in m_sel.group(1))
        hist.append((op, set(dst), sdst, half_sel))
        hist = hist[-8:]
    return errors


class State:
    def __init__(self):
        self.lgkm: list = []      # outstanding LDS ops in issue order: sets of destination registers (may be empty)
        self.vm: list = []
        self.mfma: dict = {}      # register -> instruction slots left before others may touch it

    def copy(self):
        s = State()
        s.lgkm, s.vm, s.mfma = [set(x) for x in self.lgkm], [set(x) for x in self.vm], dict(self.mfma)
        return s

    def join(self, other):
        # conservative union: keep the longer queues position by position from the young end
        def jq(a, b):
            n = max(len(a), len(b))
            a = [set()] * (n - len(a)) + a
            b = [set()] * (n - len(b)) + b
            return [x | y for x, y in zip(a, b)]
        self.lgkm, self.vm = jq(self.lgkm, other.lgkm), jq(self.vm, other.vm)
        for r, n in other.mfma.items():
            self.mfma[r] = max(self.mfma.get(r, 0), n)


def check(text: str) -> list:
    lines = [ln.strip() for ln in text.splitlines()]
    lines = [ln for ln in lines if ln and not ln.startswith(";")]
    errors = []
    # ---- 1. SCC ----
    for i, ln in enumerate(lines):
        op = ln.split()[0]
        if op in ("s_addc_u32", "s_cselect_b32") or op.startswith("s_cbranch_scc"):
            prev = lines[i - 1]
            ok = prev.split()[0].startswith(_SCC_WRITERS) and not prev.endswith(":")
            if op.startswith("s_cbranch_scc") and prev.split()[0] == "s_cselect_b32":
                ok = True
            if not ok:
                errors.append(f"line {i}: SCC consumer '{ln}' does not directly follow its producer ('{prev}')")
    errors += check_short_distances(lines)
    # ---- 2 / 3. outstanding loads and MFMA results ----
    at_label: dict = {}
    st = State()
    reachable = True              # False directly behind an unconditional branch
    for i, ln in enumerate(lines):
        if ln.endswith(":"):
            lab = ln[:-1]
            snap = at_label.get(lab)
            if not reachable:
                st = snap.copy() if snap is not None else State()
            elif snap is not None:
                st.join(snap)
            reachable = True
            continue
        op, dst, src, kind = parse(ln)
        if op in ("s_branch",) or op.startswith("s_cbranch"):
            target = ln.split()[1]
            if target in at_label:
                at_label[target].join(st)
            else:
                at_label[target] = st.copy()
            if op == "s_branch":
                reachable = False
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ln)
            if m:
                n = int(m.group(1))
                st.lgkm = st.lgkm[len(st.lgkm) - n:] if n else []
            m = re.search(r"vmcnt\((\d+)\)", ln)
            if m:
                n = int(m.group(1))
                st.vm = st.vm[len(st.vm) - n:] if n else []
            for r in st.mfma:
                st.mfma[r] -= 1
            continue
        touched = dst | src
        for q, name in ((st.lgkm, "LDS"), (st.vm, "global")):
            for pend in q:
                hit = touched & pend
                if hit:
                    errors.append(f"line {i}: '{ln}' touches {sorted(hit)[:4]} while a {name} load into them is outstanding")
        if kind == "mfma":
            # an MFMA may accumulate into (or chain from) registers a previous MFMA wrote: the hardware interlocks
            busy = {r for r in (src - dst) if st.mfma.get(r, 0) > 0 and r not in dst}
            c_regs = regs_of(split_operands(ln.split(None, 1)[1])[3])
            busy -= c_regs
        else:
            busy = {r for r in touched if st.mfma.get(r, 0) > 0}
        if busy:
            errors.append(f"line {i}: '{ln}' touches MFMA results {sorted(busy)[:4]} too early")
        slots = 1
        if op == "s_nop":
            slots = int(ln.split()[1]) + 1
        for r in list(st.mfma):
            st.mfma[r] -= slots
            if st.mfma[r] <= 0:
                del st.mfma[r]
        if kind == "mfma":
            for r in dst:
                st.mfma[r] = MFMA_WAIT
        if kind == "lds_load":
            st.lgkm.append(set(dst))
        elif kind == "lds_other":
            st.lgkm.append(set())
        elif kind == "vm_load":
            st.vm.append(set(dst))
        elif kind == "vm_other":
            st.vm.append(set())
        if op == "s_barrier":
            pass
    return errors


def load_inc(path: str) -> str:
    out = []
    for ln in open(path):
        ln = ln.rstrip("\n")
        if ln.startswith('"') and ln.endswith('\\n"'):
            out.append(ln[1:-3])
    return "\n".join(out)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "nano-vllm_amd", "csrc", "attn_prefill_loop.inc")
    errs = check(load_inc(path))
    for e in errs[:40]:
        print(e)
    print(f"{len(errs)} problem(s)")
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
