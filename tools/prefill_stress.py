#!/usr/bin/env python3
"""Development aid: the hand-scheduled prefill kernel against the compiler-scheduled one on the launch
shapes of the end-to-end run (Qwen3-0.6B heads 16/8, 8-10 ragged sequences per launch, q a strided slice
of a fused qkv buffer), many launches.  The output tensor sits inside a larger buffer whose rows behind
and in front of it are canaries, so rows written outside the sequences are detected instead of landing in
someone else's memory."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanovllm_amd import _lib, ops  # noqa: E402

SHAPES = [
    ([256] * 4 + [416] + [512] * 5, [512] * 4 + [416] + [512] * 5),
    ([96, 311] + [512] * 7, [512, 311] + [512] * 7),
    ([201, 207] + [512] * 7, [512, 207] + [512] * 7),
    ([103, 305] + [512] * 7, [103, 512] + [512] * 7),
    ([409, 511] + [512] * 6, [512, 511] + [512] * 6),
    ([408] + [512] * 7, [408] + [512] * 7),
    ([8, 401] + [512] * 7, [512, 401] + [512] * 7),
    ([5, 311] + [512] * 7, [5, 512] + [512] * 7),
    ([15, 394] + [512] * 7, [512, 394] + [512] * 7),
    ([1, 64, 65] + [512] * 7, [512, 64, 65] + [512] * 7),
]


# one or two sequences whose launches have fewer workgroups than CUs: the key range is split, unnormalised states
# cross a workspace and a merge launch writes the rows (32 / 8 heads: the chunks of one long prompt)
SPLIT_SHAPES = [
    ([512], [8192]), ([512], [4096]), ([512], [1024]), ([300], [2049]), ([512], [1536]), ([64], [5000]),
    ([129], [3000]), ([512, 1], [2560, 77]), ([257], [2048]), ([512], [7777]),
]


def run(rounds=30, H_q=16, H_kv=8, verbose=True, shapes=None, poison_workspace=False):
    """Returns (launch pairs, launch pairs whose results or canaries were wrong).  poison_workspace fills the
    split-key scratch with NaN bit patterns in front of every launch: a state that is read without having been
    written by this launch then shows as NaN rows."""
    SHAPES = shapes if shapes is not None else globals()["SHAPES"]
    dev = torch.device("cuda", 0)
    D, bs = 128, 256
    g = torch.Generator(device="cpu").manual_seed(0)
    guard = 512
    bad = 0
    launches = 0
    for rnd in range(rounds):
        for n_qs, n_ks in SHAPES:
            B = len(n_qs)
            T = sum(n_qs)
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
            lead = 3 if rnd % 2 else 0                       # decode rows in front: cu_q does not start at 0
            qkv = torch.randn(T + lead, (H_q + 2 * H_kv) * D, device=dev).bfloat16()
            q = qkv[:, :H_q * D].view(T + lead, H_q, D)
            cu_q = torch.tensor([lead] + (lead + torch.tensor(n_qs).cumsum(0)).tolist(), dtype=torch.int32, device=dev)
            cu_k = torch.tensor([0] + torch.tensor(n_ks).cumsum(0).tolist(), dtype=torch.int32, device=dev)
            outs = []
            for kern in (0, 1):
                big = torch.full((guard + T + lead + guard, H_q, D), float("nan"), dtype=torch.bfloat16, device=dev)
                o = big[guard:guard + T + lead]
                _lib.set_tuning(prefill_kernel=kern)
                if poison_workspace:
                    with torch.inference_mode():             # the buffers may have been made under it
                        for ws_ in ops._workspaces.values():
                            ws_.fill_(0xFF)
                ops.attn_prefill_varlen(q, kc, vc, cu_q, cu_k, bt, max(n_qs), max(n_ks), D ** -0.5, out=o)
                outs.append(big)
            _lib.set_tuning()
            torch.cuda.synchronize()
            other_path = os.environ.get("NV_STRESS_OTHER_LIB")
            if other_path:
                # development aid: the same launch through another build of the library, compared bit for bit
                import ctypes
                from tools.pf_two_libs import bind
                other = bind(other_path)
                t = _lib.Tuning(**{**_lib.Tuning.DEFAULTS, "prefill_kernel": 1})
                assert other.nv_set_tuning(ctypes.byref(t)) == 0
                big = torch.full((guard + T + lead + guard, H_q, D), float("nan"), dtype=torch.bfloat16, device=dev)
                o = big[guard:guard + T + lead]
                ws = torch.empty(max(16, other.nv_attn_prefill_workspace_bytes(B, T, max(n_qs), max(n_ks), H_q, H_kv, D)),
                                 dtype=torch.uint8, device=dev)
                rc = other.nv_attn_prefill_varlen(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o.data_ptr(), cu_q.data_ptr(),
                                                  cu_k.data_ptr(), bt.data_ptr(), B, width, max(n_qs), max(n_ks), H_q, H_kv, D,
                                                  nb, bs, D ** -0.5, q.stride(0), o.stride(0), 0, T + lead, ws.data_ptr(),
                                                  ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
                assert rc == 0, rc
                torch.cuda.synchronize()
                ne = outs[1][guard + lead:guard + lead + T].view(torch.int16) != big[guard + lead:guard + lead + T].view(torch.int16)
                if ne.any():
                    i = ne.nonzero()[0].tolist()
                    print(f"round {rnd} shape {n_qs}: builds differ in {int(ne.sum())} elements, first {i}: "
                          f"{outs[1][guard + lead:][tuple(i)].item()} vs {big[guard + lead:][tuple(i)].item()} "
                          f"(kernel 0: {outs[0][guard + lead:][tuple(i)].item()})", flush=True)
            launches += 1
            ref, new = outs
            canary_ok = bool(torch.isnan(new[:guard + lead]).all() and torch.isnan(new[guard + lead + T:]).all())
            body_r, body_n = ref[guard + lead:guard + lead + T].float(), new[guard + lead:guard + lead + T].float()
            err = (body_r - body_n).abs().max().item()
            nan_rows = int(torch.isnan(body_n).any(dim=(1, 2)).sum())
            # 2e-2 absolute, or one bf16 unit in the last place where that is larger (|value| >= 4: rows that see
            # one or two keys reproduce V values of that size, and the two kernels round p differently)
            over = (body_r - body_n).abs() > torch.clamp(body_r.abs() * 2.0 ** -7, min=2e-2)
            if not canary_ok or bool(over.any()) or nan_rows:
                bad += 1
                rows = (over | torch.isnan(body_n)).any(dim=(1, 2)).nonzero().flatten().tolist()
                w = (body_r - body_n).abs().flatten().argmax().item()
                print(f"  largest difference: kernel 0 {body_r.flatten()[w].item()} kernel 1 {body_n.flatten()[w].item()}")
                print(f"round {rnd} shape {n_qs}: canaries {'ok' if canary_ok else 'TOUCHED'} max diff {err:.4g} "
                      f"nan rows {nan_rows} bad rows {rows[:12]} (total {len(rows)})", flush=True)
        if verbose and rnd % 10 == 9:
            print(f"round {rnd + 1}: {launches} launch pairs, {bad} bad", flush=True)
    return launches, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--hq", type=int, default=16)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--split", action="store_true", help="the key-range split shapes (32 / 8 heads) with a poisoned workspace")
    args = ap.parse_args()
    if args.split:
        launches, bad = run(args.rounds, 32, 8, shapes=SPLIT_SHAPES, poison_workspace=True)
    else:
        launches, bad = run(args.rounds, args.hq, args.hkv)
    print(f"done: {launches} launch pairs, {bad} bad")


if __name__ == "__main__":
    main()
