#!/usr/bin/env python3
"""Randomised differential run of the attention kernels against the oracle (development aid; the fixed cases live
in tests/): random head counts, block sizes, batch compositions (decode rows, chunks, whole prompts, empty and
one-token sequences), lengths, softmax scales and query strides; both prefill kernels, the launch plan's choice,
forced key-range splits, both decode work-item forms and combines, decode launches with the host's chunk.  Output rows sit inside NaN guard rows.

    python tests/fuzz_attention.py --iters 300 [--seed 0]
"""
import argparse
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_paged_problem, row_bounds  # noqa: E402
from oracle import ref_attention  # noqa: E402  (the checker, as in tests/)
from nanovllm_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
HEADS = [(16, 8), (32, 8), (8, 8), (8, 2), (8, 1), (64, 8), (6, 2), (4, 4), (12, 4), (16, 2)]


def one(seed, verbose):
    r = random.Random(seed)
    H_q, H_kv = r.choice(HEADS)
    bs = r.choice([64, 128, 256, 256, 256, 512])
    B = r.randint(1, 6)
    n_qs, n_ks = [], []
    for _ in range(B):
        kind = r.choice(["decode", "decode", "chunk", "chunk", "full", "one", "long_chunk"])
        if kind == "decode":
            n_k = r.choice([1, 2, r.randint(1, 600), r.randint(1, 5000)]); n_q = 1
        elif kind == "chunk":
            n_k = r.randint(2, 3000); n_q = r.randint(1, min(n_k, 600))
        elif kind == "long_chunk":
            n_k = r.randint(1500, 9000); n_q = r.choice([64, 128, 256, 512, r.randint(1, 512)])
        elif kind == "full":
            n_k = r.randint(1, 1500); n_q = n_k
        else:
            n_k = 1; n_q = 1
        n_qs.append(n_q); n_ks.append(n_k)
    decode_only = all(n == 1 for n in n_qs)
    nb = sum((n + bs - 1) // bs for n in n_ks) + r.randint(1, 3)
    p = make_paged_problem(seed, n_qs, n_ks, H_q, H_kv, bs, nb)
    p["scale"] = r.choice([128 ** -0.5, 128 ** -0.5, 128 ** -0.5 * 4, 0.02, 0.3])
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"], p["block_tables"],
                                            p["scale"])
    bound = torch.maximum(row_bounds(ref, 0.05), torch.tensor(4e-3))
    d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in p.items()}
    if r.random() < 0.5:      # q as a view into a fused projection output
        T = d["q"].shape[0]
        wide = torch.zeros(T, (H_q + 2 * H_kv) * 128, dtype=torch.bfloat16, device=DEV)
        wide[:, :H_q * 128] = d["q"].reshape(T, -1)
        d["q"] = wide[:, :H_q * 128].view(T, H_q, 128)
    variants = []
    if decode_only:
        for form in (0, 1):
            for single in (0, 1):
                variants.append((dict(decode_work_item=form, decode_single_launch=single), "decode"))
        # the host's choice of keys per work item (rotated grid, direct rows next to split ones)
        for form in (-1, 0, 1):
            variants.append((dict(decode_work_item=form), ("decode", r.choice([64, 128, 256, 320, 512, 1000, 2048]))))
    else:
        for kern in (-1, 0, 1):
            variants.append((dict(prefill_kernel=kern), "prefill"))
        variants.append((dict(prefill_kernel=r.choice([0, 1]), prefill_split=r.choice([2, 3, 5])), "prefill"))
        variants.append((dict(), "mixed"))
    bad = 0
    T = d["q"].shape[0]
    for tune, mode in variants:
        _lib.set_tuning(**tune)
        guard = 8
        big = torch.full((guard + T + guard, H_q, 128), float("nan"), dtype=torch.bfloat16, device=DEV)
        o = big[guard:guard + T]
        if isinstance(mode, tuple):
            ops.attn_decode_paged(d["q"], d["k_cache"], d["v_cache"], d["cu_k"], d["block_tables"], p["max_k"], p["scale"],
                                  cu_seqlens_q=d["cu_q"], out=o, chunk_keys=mode[1])
        elif mode == "decode":
            ops.attn_decode_paged(d["q"], d["k_cache"], d["v_cache"], d["cu_k"], d["block_tables"], p["max_k"], p["scale"],
                                  cu_seqlens_q=d["cu_q"], out=o)
        elif mode == "prefill":
            ops.attn_prefill_varlen(d["q"], d["k_cache"], d["v_cache"], d["cu_q"], d["cu_k"], d["block_tables"], p["max_q"],
                                    p["max_k"], p["scale"], out=o)
        else:
            ops.attn_prefill_varlen(d["q"], d["k_cache"], d["v_cache"], d["cu_q"], d["cu_k"], d["block_tables"], p["max_q"],
                                    p["max_k"], p["scale"], skip_single_query=True, out=o)
            ops.attn_decode_paged(d["q"], d["k_cache"], d["v_cache"], d["cu_k"], d["block_tables"], p["max_k"], p["scale"],
                                  cu_seqlens_q=d["cu_q"], out=o)
        torch.cuda.synchronize()
        _lib.set_tuning()
        oc = big.float().cpu()
        guards_ok = bool(torch.isnan(oc[:guard]).all() and torch.isnan(oc[guard + T:]).all())
        err = (oc[guard:guard + T] - ref).abs().amax(dim=-1)
        ok = guards_ok and bool(torch.isfinite(oc[guard:guard + T]).all()) and bool((err <= bound).all())
        if not ok:
            bad += 1
            worst = (err / bound).max().item() if torch.isfinite(err).all() else float("nan")
            print(f"BAD seed {seed} H {H_q}/{H_kv} bs {bs} n_q {n_qs} n_k {n_ks} scale {p['scale']:.4f} {tune} {mode}: "
                  f"guards {guards_ok}, worst err/bound {worst:.2f}", flush=True)
    if verbose:
        print(f"seed {seed}: H {H_q}/{H_kv} bs {bs} n_q {n_qs} n_k {n_ks}: {len(variants)} variants, {bad} bad", flush=True)
    return len(variants), bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    n = bad = 0
    for i in range(args.iters):
        a, b = one(args.seed + i, args.verbose)
        n += a; bad += b
        if (i + 1) % 20 == 0:
            print(f"{i + 1} problems, {n} launches, {bad} bad", flush=True)
    print(f"done: {args.iters} problems, {n} launches, {bad} bad")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
