"""Bandwidth kernels of the hot path other than attention: time per launch, algorithmic GB/s, % of 8 TB/s.

    python tools/small_kernels_bench.py [--iters 200] [--json out.json]

Sizes: T = 256 (one decode step of the headline workload), T = 4000 (one prefill step under the
reference's token budget) and T = 65536 (asymptote).  Inputs rotate over enough buffer sets to exceed
the 256 MB last-level cache when a single set would fit.  Algorithmic bytes:
  store            T * (2 * H_kv * D * 2) * 2 + 4 T
  qk-norm+rope+store  2 * T * (H_q + 2 H_kv) * D * 2 + T * D * 4 + 12 T
  rope (unfused)   2 * T * (H_q + H_kv) * D * 2 + T * D * 4 + 8 T
  rmsnorm          rows * cols * 2 * 2 + cols * 2
  add + rmsnorm    rows * cols * 2 * 4 + cols * 2
  silu-and-mul     rows * inter * 2 * 3
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("nano-vllm_amd.ops")

PEAK = 8000.0  # GB/s
LLC = 256 << 20


def _time(fn_sets, iters):
    """Average device time of one launch: `iters` launches captured in one HIP graph (so that the host's
    launch rate does not bound the small cases), replayed three times."""
    n = len(fn_sets)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(min(3, iters)):
            fn_sets[i % n]()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn_sets[i % n]()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (3 * iters)
    del g
    return us


def _nsets(bytes_per_set):
    return max(1, min(16, -(-2 * LLC // max(bytes_per_set, 1))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    H_q, H_kv, D, bs = 16, 8, 128, 256
    rows_out = []

    def report(name, shape, us, nbytes):
        gbs = nbytes / us / 1e3
        rows_out.append({"kernel": name, "shape": shape, "us": round(us, 2), "GBps": round(gbs, 1),
                         "pct_of_8TBps": round(100 * gbs / PEAK, 1), "bytes": nbytes})
        print(f"{name:>22s} {shape:>28s}: {us:9.2f} us  {gbs:8.1f} GB/s  ({100 * gbs / PEAK:5.1f} %)", flush=True)

    for T in (256, 4000, 65536):
        NB = -(-T // bs) + 1
        per_set = T * (H_q + 2 * H_kv) * D * 2 * 2
        n = _nsets(per_set)
        sets = []
        for _ in range(n):
            qkv = torch.randn(T, (H_q + 2 * H_kv) * D, device=dev, dtype=torch.bfloat16)
            q = qkv[:, :H_q * D].view(T, H_q, D)
            k = qkv[:, H_q * D:(H_q + H_kv) * D].view(T, H_kv, D)
            v = qkv[:, (H_q + H_kv) * D:].view(T, H_kv, D)
            kc = torch.zeros(NB, bs, H_kv, D, device=dev, dtype=torch.bfloat16)
            vc = torch.zeros_like(kc)
            slots = torch.randperm(NB * bs, device=dev)[:T].to(torch.int32)
            pos = torch.randint(0, 40000, (T,), device=dev, dtype=torch.int64)
            sets.append((q, k, v, kc, vc, slots, pos))
        qw = torch.randn(D, device=dev, dtype=torch.bfloat16)
        kw = torch.randn(D, device=dev, dtype=torch.bfloat16)
        half = D // 2
        inv = 1.0 / (1e6 ** (torch.arange(0, half, device=dev, dtype=torch.float32) * 2 / D))
        fr = torch.arange(40960, device=dev, dtype=torch.float32)[:, None] * inv[None]
        cs = torch.cat([fr.cos(), fr.sin()], -1).contiguous()
        shp = f"T={T} H={H_q}/{H_kv}"
        us = _time([lambda s=s: ops.store_kvcache(s[1], s[2], s[3], s[4], s[5]) for s in sets], args.iters)
        report("store_kvcache", shp, us, T * (2 * H_kv * D * 2) * 2 + 4 * T)
        us = _time([lambda s=s: ops.qknorm_rope_store(s[6], s[0], s[1], s[2], qw, kw, cs, 1e-6, s[5], s[3], s[4],
                                                       want_k=False) for s in sets], args.iters)
        report("qknorm_rope_store", shp, us, 2 * T * (H_q + 2 * H_kv) * D * 2 + T * D * 4 + 12 * T)
        us = _time([lambda s=s: ops.rope(s[6], s[0], s[1], cs) for s in sets], args.iters)
        report("rope", shp, us, 2 * T * (H_q + H_kv) * D * 2 + T * D * 4 + 8 * T)
        del sets

        for cols in (1024, 4096):
            n = _nsets(T * cols * 2 * 4)
            xs = [(torch.randn(T, cols, device=dev, dtype=torch.bfloat16),
                   torch.randn(T, cols, device=dev, dtype=torch.bfloat16)) for _ in range(n)]
            w = torch.randn(cols, device=dev, dtype=torch.bfloat16)
            us = _time([lambda x=x: ops.rmsnorm(x[0], w, 1e-6) for x in xs], args.iters)
            report("rmsnorm", f"{T}x{cols}", us, T * cols * 2 * 2 + cols * 2)
            us = _time([lambda x=x: ops.add_rmsnorm(x[0], x[1], w, 1e-6) for x in xs], args.iters)
            report("add_rmsnorm", f"{T}x{cols}", us, T * cols * 2 * 4 + cols * 2)
            del xs
        for inter in (3072, 12288):
            n = _nsets(T * inter * 2 * 3)
            xs = [torch.randn(T, 2 * inter, device=dev, dtype=torch.bfloat16) for _ in range(n)]
            us = _time([lambda x=x: ops.silu_and_mul(x) for x in xs], args.iters)
            report("silu_and_mul", f"{T}x{inter}", us, T * inter * 2 * 3)
            del xs
        torch.cuda.empty_cache()
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows_out, f, indent=1)


if __name__ == "__main__":
    main()
