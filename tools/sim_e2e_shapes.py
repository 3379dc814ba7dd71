#!/usr/bin/env python3
"""CPU-only: replay the scheduling of tools/e2e_bench.py's workload (no model, dummy tokens) and list the
distinct attention launches the engine would make (what Attention.attend passes to the two kernels)."""
import os
import sys
from collections import Counter

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanovllm_amd  # noqa: E402,F401
from nanovllm_amd.engine.scheduler import Scheduler  # noqa: E402
from nanovllm_amd.engine.sequence import SamplingParams, Sequence  # noqa: E402
from nanovllm_amd.engine.model_runner import prepare_inputs  # noqa: E402
from nanovllm_amd.engine.llm_engine import _is_decode_row  # noqa: E402


def launches(m):
    """(kind, n_qs, n_ks, max_q, max_k) per kernel launch, as Attention.attend decides."""
    cu_q, cu_k = m["cu_seqlens_q"], m["cu_seqlens_k"]
    nq, nk = np.diff(cu_q).tolist(), np.diff(cu_k).tolist()
    B = len(nq)
    out = []
    if m["max_seqlen_q"] <= 1:
        return [("decode", nq, nk, 1, m["max_seqlen_k"])]
    nd = m["num_leading_decode"]
    if nd is not None:
        if nd > 0:
            out.append(("decode", nq[:nd], nk[:nd], 1, m["max_seqlen_k_decode"]))
        if nd < B:
            out.append(("prefill", nq[nd:], nk[nd:], m["max_seqlen_q_rest"], m["max_seqlen_k_rest"]))
        return out
    n_dec = m["num_decode_seqs"]
    if n_dec is not None and n_dec == B:
        return [("decode", nq, nk, 1, m["max_seqlen_k"])]
    mixed = n_dec is None or n_dec > 0
    out.append(("prefill-skip1" if mixed else "prefill", nq, nk, m["max_seqlen_q"], m["max_seqlen_k"]))
    if mixed:
        out.append(("decode-skipmulti", nq, nk, 1, m["max_seqlen_k"]))
    return out


def prefill_kernel_choice(B, H_q, H_kv, max_q, max_k):
    """The launch plan of nv_attn_prefill_varlen (csrc/attn_prefill.hip, prefill_shape), heuristic mode."""
    from math import gcd
    G = H_q // H_kv
    hpw = gcd(G, 4)
    rows_per_wg = 64 * (4 // hpw)
    n_qtiles = -(-max_q // rows_per_wg)
    big_wgs = B * H_kv * n_qtiles * (G // hpw)
    tiles = -(-max_k // 64)
    return "hand-scheduled" if (big_wgs >= 256 and tiles >= 16 and tiles <= 1024) else "compiler-scheduled"


def run(batch=256, prompt_len=512, out_len=256, budget=4000, bs=256):
    blocks = batch * (-(-(prompt_len + out_len) // bs)) + 8
    sch = Scheduler(blocks, bs, budget, 512, -1, None)
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(0, 151936, (prompt_len,), generator=g).tolist() for _ in range(batch)]
    warm = [torch.randint(0, 151936, (prompt_len,), generator=g).tolist() for _ in range(batch)]
    seen = Counter()
    examples = {}
    for ps, n_out in ((warm[:4], 4), (warm, 2), (prompts, out_len)):
        for p in ps:
            sch.add(Sequence(p, SamplingParams(0.0, n_out)))
        while not sch.is_finished():
            seqs, _ = sch.schedule()
            assert seqs
            m = prepare_inputs(seqs, bs)
            for L in launches(m):
                kind, nq, nk, mq, mk = L
                key = (kind, len(nq), mq, mk, tuple(sorted(Counter(zip(nq, nk)).items()))[:6])
                seen[key] += 1
                examples.setdefault(key, (nq, nk, m["block_tables"].shape, int(m["block_tables"].min()),
                                          int(m["block_tables"].max())))
            toks = [7 if s.completes_step else None for s in seqs]
            sch.update_from_output(seqs, toks)
    for key, n in seen.items():
        kind, B, mq, mk, comp = key
        if kind.startswith("prefill") or n < 3:
            which = prefill_kernel_choice(B, 16, 8, mq, mk) if kind.startswith("prefill") else ""
            print(f"{n:5d} x {kind:16s} B={B:3d} max_q={mq:4d} max_k={mk:5d} {comp}  tbl={examples[key][2:]} {which}")
    print("decode launch kinds:", sum(1 for k in seen if k[0].startswith("decode")))


if __name__ == "__main__":
    run()
