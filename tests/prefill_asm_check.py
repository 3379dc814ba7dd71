#!/usr/bin/env python3
"""Development aid: the hand-scheduled prefill kernel (forced through nv_set_tuning) against the CPU oracle on
a ladder of shapes, smallest first; stops at the first mismatch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_paged_problem  # noqa: E402
from nanovllm_amd import _lib, ops  # noqa: E402
from oracle import ref_attention  # noqa: E402

CASES = [
    # n_qs, n_ks, H_q, H_kv
    ([64], [64], 4, 1),
    ([64], [128], 4, 1),
    ([128], [128], 4, 1),
    ([64], [192], 4, 1),
    ([256], [256], 4, 1),
    ([512], [512], 32, 8),
    ([70], [70], 4, 1),
    ([5], [13], 8, 2),
    ([1, 31, 32, 33, 127, 128, 129], [1, 31, 32, 33, 127, 128, 129], 16, 8),
    ([512, 512, 300], [1024, 512, 812], 16, 8),
    ([64, 1, 200, 1], [64, 300, 456, 1], 32, 8),
    ([300], [2049], 4, 4),
    ([3], [1], 4, 2),
    ([2048], [2048], 8, 1),
    # two query tiles per workgroup (>= 512 workgroups): even and odd tile counts, ragged, chunked
    ([1024] * 16, [1024] * 16, 16, 2),
    ([960, 1000, 64, 70] * 4, [960, 1300, 64, 326] * 4, 16, 2),
]


def main():
    dev = "cuda:0"
    only = [int(x) for x in sys.argv[1:]]
    for ci, (n_qs, n_ks, H_q, H_kv) in enumerate(CASES):
        if only and ci not in only:
            continue
        bs = 256
        nb = sum((n + bs - 1) // bs for n in n_ks) + 1
        p = make_paged_problem(ci + 1, n_qs, n_ks, H_q, H_kv, bs, nb)
        ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                                p["block_tables"], p["scale"])
        pd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in p.items()}
        _lib.set_tuning(prefill_kernel=1)
        print(f"case {ci}: n_q {n_qs} n_k {n_ks} H {H_q}/{H_kv} ...", flush=True)
        o = ops.attn_prefill_varlen(pd["q"], pd["k_cache"], pd["v_cache"], pd["cu_q"], pd["cu_k"], pd["block_tables"],
                                    p["max_q"], p["max_k"], p["scale"])
        torch.cuda.synchronize()
        _lib.set_tuning()
        err = (o.float().cpu() - ref).abs()
        bad = torch.isnan(o.float().cpu()).sum().item()
        print(f"   max err {err.max().item():.4g}  nan {bad}", flush=True)
        if not (err.max().item() <= 2e-2) or bad:
            rows = (err.amax(dim=(1, 2)) > 2e-2).nonzero().flatten().tolist()
            print("   bad rows:", rows[:40], "... total", len(rows))
            heads = (err.amax(dim=(0, 2)) > 2e-2).nonzero().flatten().tolist()
            dims = (err.amax(dim=(0, 1)) > 2e-2).nonzero().flatten().tolist()
            print("   bad heads:", heads[:16], " bad dims:", dims[:32], "total dims", len(dims))
            r0 = rows[0] if rows else 0
            print("   got ", o[r0, 0, :8].float().cpu().tolist())
            print("   want", ref[r0, 0, :8].tolist())
            sys.exit(1)
    print("all ok")


if __name__ == "__main__":
    main()
