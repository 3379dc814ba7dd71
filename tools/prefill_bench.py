#!/usr/bin/env python3
"""Tuning aid: time nv_attn_prefill_varlen on a few prefill / chunked-prefill shapes and report
algorithmic TFLOP/s (visible (query,key) pairs only, 4*D flops per pair and query head)."""
import argparse
import math
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanovllm_amd import _lib  # noqa: E402

CASES = {
    # name: (n_qs, n_ks, H_q, H_kv)
    "cfg2_prefill_7x512": ([512] * 7, [512] * 7, 16, 8),
    "cfg3_chunk_last_8k": ([512], [8192], 32, 8),
    "cfg3_chunk_mid_4k": ([512], [4096], 32, 8),
    "cfg3_7seq_chunks": ([512] * 7, [512, 1024, 2048, 4096, 8192, 6144, 3072], 32, 8),
    # the same step with its sequences ordered by key count (what a runner that holds the lengths could do)
    "cfg3_7seq_longest_first": ([512] * 7, [8192, 6144, 4096, 3072, 2048, 1024, 512], 32, 8),
    "cfg3_7seq_shortest_first": ([512] * 7, [512, 1024, 2048, 3072, 4096, 6144, 8192], 32, 8),
    # uneven whole-prompt prefill (8 prompts of 256 .. 8192 tokens) in three orders
    "uneven_full": ([1024, 256, 8192, 512, 4096, 2048, 512, 3072], [1024, 256, 8192, 512, 4096, 2048, 512, 3072], 32, 8),
    "uneven_full_longest_first": ([8192, 4096, 3072, 2048, 1024, 512, 512, 256], [8192, 4096, 3072, 2048, 1024, 512, 512, 256], 32, 8),
    "uneven_full_shortest_first": ([256, 512, 512, 1024, 2048, 3072, 4096, 8192], [256, 512, 512, 1024, 2048, 3072, 4096, 8192], 32, 8),
    "b16_n2048_h32": ([2048] * 16, [2048] * 16, 32, 8),
    "b4_n8192_h32": ([8192] * 4, [8192] * 4, 32, 8),
    "b16_n2048_h64": ([2048] * 16, [2048] * 16, 64, 8),
    # head shards of Qwen3-32B (64 / 8 heads over 1, 2, 4, 8 GPUs): one sequence, a 512-token chunk over 4096 keys, and a
    # whole 8192-token prompt
    "tp1_chunk": ([512], [4096], 64, 8), "tp2_chunk": ([512], [4096], 32, 4),
    "tp4_chunk": ([512], [4096], 16, 2), "tp8_chunk": ([512], [4096], 8, 1),
    "tp1_full": ([8192], [8192], 64, 8), "tp2_full": ([8192], [8192], 32, 4),
    "tp4_full": ([8192], [8192], 16, 2), "tp8_full": ([8192], [8192], 8, 1),
    # pair counts that do not fill the 8 XCDs evenly: 3 sequences x 4 kv heads, 7 x 1, 3 x 8
    "pairs12_full": ([4096] * 3, [4096] * 3, 32, 4), "pairs7_full": ([4096] * 7, [4096] * 7, 8, 1),
    "pairs24_full": ([4096] * 3, [4096] * 3, 32, 8),
    # per-workgroup cost probes: 4096 workgroups of the hand-scheduled kernel with 1 / 2 / 4 tiles each
    "wg_probe_1tile": ([64] * 512, [64] * 512, 32, 8),
    "wg_probe_2tile": ([64] * 512, [128] * 512, 32, 8),
    "wg_probe_4tile": ([64] * 512, [256] * 512, 32, 8),
    "wg_probe_8tile": ([64] * 512, [512] * 512, 32, 8),
}


def flops(n_qs, n_ks, H_q, D=128):
    return sum(4 * D * H_q * (nq * (nk - nq) + nq * (nq + 1) // 2) for nq, nk in zip(n_qs, n_ks))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--lib", default="")
    ap.add_argument("--share-kv", action="store_true", help="every sequence uses the first sequence's blocks (small K/V footprint)")
    ap.add_argument("--split", type=int, default=0, help="nv_tuning.prefill_split (0 = the launch plan's choice)")
    ap.add_argument("--kernels", default="auto", help="comma list of auto,0,1 (nv_tuning.prefill_kernel: 0 compiler-scheduled, 1 hand-scheduled)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    dll = _lib.load(args.lib or None)
    stream = torch.cuda.current_stream(dev).cuda_stream
    bs, D = 256, 128
    for name in args.cases.split(","):
        if name.startswith("full_"):                  # full_<sequences>x<tokens>: whole-prompt prefill, 32 / 8 heads
            nb_, nt_ = (int(x) for x in name[5:].split("x"))
            CASES[name] = ([nt_] * nb_, [nt_] * nb_, 32, 8)
        if name.startswith("chunk_"):                 # chunk_<queries>x<keys>: one sequence, 32 / 8 heads
            nq, nk = (int(x) for x in name[6:].split("x"))
            CASES[name] = ([nq], [nk], 32, 8)
        n_qs, n_ks, H_q, H_kv = CASES[name]
        B = len(n_qs)
        blocks = [math.ceil(n / bs) for n in n_ks]
        NB = sum(blocks)
        g = torch.Generator().manual_seed(0)
        perm = torch.randperm(NB, generator=g).tolist()
        width = max(blocks)
        tables, nxt = [], 0
        for nb in blocks:
            tables.append(perm[nxt:nxt + nb] + [-1] * (width - nb))
            nxt += nb
        if args.share_kv:
            tables = [tables[0][:len(t)] + [-1] * (len(t) - len(tables[0])) if blocks[i] <= blocks[0] else t
                      for i, t in enumerate(tables)]
        bt = torch.tensor(tables, dtype=torch.int32, device=dev)
        cu_q = torch.tensor([0] + torch.tensor(n_qs).cumsum(0).tolist(), dtype=torch.int32, device=dev)
        cu_k = torch.tensor([0] + torch.tensor(n_ks).cumsum(0).tolist(), dtype=torch.int32, device=dev)
        T = sum(n_qs)
        q = torch.randn(T, H_q, D, device=dev).bfloat16()
        kc = torch.randn(NB, bs, H_kv, D, device=dev).bfloat16()
        vc = torch.randn(NB, bs, H_kv, D, device=dev).bfloat16()
        o = torch.empty_like(q)

        ws = torch.empty(max(16, dll.nv_attn_prefill_workspace_bytes(B, T, max(n_qs), max(n_ks), H_q, H_kv, D)),
                         dtype=torch.uint8, device=dev)

        def launch():
            rc = dll.nv_attn_prefill_varlen(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o.data_ptr(), cu_q.data_ptr(),
                                            cu_k.data_ptr(), bt.data_ptr(), B, width, max(n_qs), max(n_ks), H_q, H_kv, D,
                                            NB, bs, D ** -0.5, q.stride(0), o.stride(0), 0, T, ws.data_ptr(), ws.numel(),
                                            stream)
            assert rc == 0, dll.nv_last_error()

        f = flops(n_qs, n_ks, H_q)
        kernels = args.kernels.split(",")
        ts = {k: [] for k in kernels}
        for rnd in range(args.rounds + 1):
            for k in kernels:                      # interleaved rounds in one process
                t = _lib.Tuning(**{**_lib.Tuning.DEFAULTS, **({} if k == "auto" else {"prefill_kernel": int(k)}),
                                   "prefill_split": args.split})
                assert dll.nv_set_tuning(t) == 0
                need = dll.nv_attn_prefill_workspace_bytes(B, T, max(n_qs), max(n_ks), H_q, H_kv, D)
                if need > ws.numel():
                    ws = torch.empty(need, dtype=torch.uint8, device=dev)
                launch()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.iters):
                    launch()
                b.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    ts[k].append(a.elapsed_time(b) / args.iters)
        dll.nv_set_tuning(None)
        for k in kernels:
            med = statistics.median(ts[k])
            print(f"{name:>22s} kernel={k:>4s}: {med*1e3:9.1f} us  {f/med/1e9:8.1f} TFLOP/s  ({100*f/med/1e9/2500:5.1f}% of 2.5 PF)  "
                  f"flops={f/1e9:.1f} G", flush=True)


if __name__ == "__main__":
    main()
