#!/usr/bin/env python3
"""Development aid: the hand-scheduled prefill kernel of TWO builds of the library in one process, on the
engine-step shapes of tools/prefill_stress.py with seeded data: outputs must be bit-identical when a change is
meant to reorder instructions only.  Prints every launch whose outputs differ and where.
    python tools/pf_two_libs.py path/to/other_lib.so [rounds]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanovllm_amd import _lib  # noqa: E402
from tools.prefill_stress import SHAPES  # noqa: E402


def bind(path):
    dll = ctypes.CDLL(path)
    for name, (res, args) in _lib.PROTOTYPES.items():
        fn = getattr(dll, name)
        fn.restype, fn.argtypes = res, args
    return dll


def main():
    other = bind(sys.argv[1])
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    base = _lib.lib()
    dev = torch.device("cuda", 0)
    H_q, H_kv, D, bs = 16, 8, 128, 256
    torch.manual_seed(1234)
    g = torch.Generator().manual_seed(0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    t = _lib.Tuning(**{**_lib.Tuning.DEFAULTS, "prefill_kernel": 1})
    for dll in (base, other):
        assert dll.nv_set_tuning(ctypes.byref(t)) == 0
    ndiff = 0
    for rnd in range(rounds):
        for si, (n_qs, n_ks) in enumerate(SHAPES):
            B, T = len(n_qs), sum(n_qs)
            need = [-(-n // bs) for n in n_ks]
            nb = sum(need) + 3
            perm = torch.randperm(nb, generator=g).tolist()
            width = max(need) + 1
            tables, nxt = [], 0
            for n in need:
                tables.append(perm[nxt:nxt + n] + [-1] * (width - n))
                nxt += n
            bt = torch.tensor(tables, dtype=torch.int32, device=dev)
            kc = torch.randn(nb, bs, H_kv, D, device=dev).bfloat16()
            vc = torch.randn(nb, bs, H_kv, D, device=dev).bfloat16()
            q = torch.randn(T, H_q, D, device=dev).bfloat16()
            cu_q = torch.tensor([0] + torch.tensor(n_qs).cumsum(0).tolist(), dtype=torch.int32, device=dev)
            cu_k = torch.tensor([0] + torch.tensor(n_ks).cumsum(0).tolist(), dtype=torch.int32, device=dev)
            outs = []
            for dll in (base, other):
                o = torch.zeros_like(q)
                ws_b = dll.nv_attn_prefill_workspace_bytes(B, T, max(n_qs), max(n_ks), H_q, H_kv, D)
                ws = torch.empty(max(16, ws_b), dtype=torch.uint8, device=dev)
                rc = dll.nv_attn_prefill_varlen(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o.data_ptr(), cu_q.data_ptr(),
                                                cu_k.data_ptr(), bt.data_ptr(), B, width, max(n_qs), max(n_ks), H_q, H_kv, D,
                                                nb, bs, D ** -0.5, q.stride(0), o.stride(0), 0, T, ws.data_ptr(), ws.numel(),
                                                stream)
                assert rc == 0, rc
                torch.cuda.synchronize()
                outs.append(o)
            a, b = outs
            ne = (a.view(torch.int16) != b.view(torch.int16))
            if ne.any():
                ndiff += 1
                rows = ne.any(dim=(1, 2)).nonzero().flatten().tolist()
                idx = ne.nonzero()[0].tolist()
                print(f"round {rnd} shape {si}: {int(ne.sum())} elements differ in rows {rows[:10]} (total {len(rows)}); "
                      f"first at {idx}: {a[tuple(idx)].item()} vs {b[tuple(idx)].item()}", flush=True)
    print(f"done: {rounds * len(SHAPES)} launches, {ndiff} with differing outputs")


if __name__ == "__main__":
    main()
