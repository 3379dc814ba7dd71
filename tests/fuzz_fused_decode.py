#!/usr/bin/env python3
"""Randomised differential run of the one-launch decode step (nv_attn_decode_paged_fused) against the two launches
it replaces (development aid; the fixed cases live in tests/test_gpu_fused_decode.py, whose helpers and bounds this
uses): random head groupings 1 .. 8, block sizes, 1 .. 24 sequences of 1 .. 40,000 keys (tile and block
boundaries favoured), with and without q/k-norm, strided projections, both work-item forms and the plan's choice.
Cache bytes must be equal; outputs may differ by one bf16 ulp on one value in a thousand.

    python tests/fuzz_fused_decode.py --iters 300 [--seed 0]
"""
import argparse
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fused_decode import _assert_same, _both, _step  # noqa: E402
from nanovllm_amd import _lib, ops  # noqa: E402

HEADS = [(16, 8), (8, 8), (2, 1), (32, 8), (64, 8), (8, 1), (24, 8), (40, 8), (7, 1), (12, 2), (6, 1), (16, 2), (32, 4)]


def one(seed, verbose):
    r = random.Random(seed)
    H_q, H_kv = r.choice(HEADS)
    bs = r.choice([16, 32, 64, 128, 256, 256, 256, 512])
    B = r.choice([1, 1, 2, 3, 5, 8, 13, 24])
    budget = 60000                       # keys per problem (the caches are cloned twice)
    ctx = []
    for _ in range(B):
        kind = r.choice(["tiny", "edge", "short", "mid", "long"])
        if kind == "tiny":
            n = r.randint(1, 3)
        elif kind == "edge":
            n = max(1, r.choice([16, bs, 2 * bs, 3 * bs]) * r.randint(1, 3) + r.randint(-1, 1))
        elif kind == "short":
            n = r.randint(1, 800)
        elif kind == "mid":
            n = r.randint(800, 6000)
        else:
            n = r.randint(6000, 40000)
        n = max(1, min(n, budget - (B - len(ctx))))
        budget -= n
        ctx.append(n)
    norm = r.random() < 0.6
    strided = r.random() < 0.6
    st = _step(seed, ctx, H_q, H_kv, bs, norm, strided=strided)
    bad = 0
    forms = [-1, 0, 1]
    for wi in forms:
        _lib.set_tuning(decode_work_item=wi)
        try:
            a, b = _both(ops, *st)
            _assert_same(a, b, f"seed {seed}")
        except AssertionError as e:
            bad += 1
            print(f"BAD seed {seed} H {H_q}/{H_kv} bs {bs} norm {norm} strided {strided} work item {wi} ctx {ctx}: {e}",
                  flush=True)
        finally:
            _lib.set_tuning()
    if verbose:
        print(f"seed {seed}: H {H_q}/{H_kv} bs {bs} norm {norm} ctx {ctx}: {bad} bad", flush=True)
    return len(forms), bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    n = bad = 0
    for i in range(args.iters):
        a, b = one(args.seed + i, args.verbose)
        n += a
        bad += b
        if (i + 1) % 20 == 0:
            print(f"{i + 1} problems, {n} launch pairs, {bad} bad", flush=True)
    print(f"done: {args.iters} problems, {n} launch pairs, {bad} bad")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
