#!/usr/bin/env python3
"""Development aid: one live 64-row wave over two key tiles; compares the hand-scheduled kernel's rows 48..63
with what specific stale-register hypotheses would produce."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_paged_problem  # noqa: E402
from nanovllm_amd import _lib, ops  # noqa: E402

p = make_paged_problem(2, [64], [128], 4, 1, 256, 2)
pd = {k: (v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in p.items()}
_lib.set_tuning(prefill_kernel=1)
for rep in range(3):
    o = ops.attn_prefill_varlen(pd["q"], pd["k_cache"], pd["v_cache"], pd["cu_q"], pd["cu_k"], pd["block_tables"],
                                p["max_q"], p["max_k"], p["scale"])
    torch.cuda.synchronize()
    o = o.double().cpu()
    blk = int(p["block_tables"][0, 0])
    K = p["k_cache"][blk, :128, 0].double(); V = p["v_cache"][blk, :128, 0].double()
    for h in range(4):
        q = p["q"][:, h].double()
        s = (q @ K.T) * p["scale"]
        vis = torch.arange(128)[None, :] <= (64 + torch.arange(64))[:, None]
        s = s.masked_fill(~vis, float("-inf"))
        m = s.amax(1, keepdim=True)
        pr = torch.exp(s - m)
        l = pr.sum(1, keepdim=True)
        ref = pr @ V / l
        cands = {"ref": ref}
        pA = pr.clone(); pA[:, 112:] = 0; cands["drop p(112:)"] = pA @ V / l
        VB = V.clone(); VB[112:] = V[48:64]; cands["V(112:) <- V(48:64)"] = pr @ VB / l
        pC = pr.clone(); pC[:, 112:] = pr[:, 48:64]; cands["p(112:) <- p(48:64)"] = pC @ V / l
        m0 = s[:, :64].amax(1, keepdim=True)
        pD = torch.exp(s - m0); cands["m from tile 0 only (same result)"] = pD @ V / pD.sum(1, keepdim=True)
        VE = V.clone(); VE[112:] = 0; cands["V(112:) = 0"] = pr @ VE / l
        rows = slice(48, 64)
        print(f"rep {rep} head {h}: " + "  ".join(f"[{k}] {(o[rows, h] - c[rows]).abs().max():.3g}" for k, c in cands.items()), flush=True)
        err = (o[:, h] - ref).abs().amax(1)
        if err.max() > 0.02:
            w = o[:, h] @ torch.linalg.inv(V)
            wref = pr / l
            d = (w - wref)
            for r in (48, 52, 55, 59, 63):
                ks = (d[r].abs() > 0.004).nonzero().flatten().tolist()
                print(f"   row {r}: keys off {ks}")
                print("      w   ", [round(float(w[r, k]), 4) for k in ks[:12]])
                print("      wref", [round(float(wref[r, k]), 4) for k in ks[:12]])
                lk = float(pD[r].sum()); pe = float(w[r, 68]) * lk
                print(f"      key 68: p_eff {pe:.4f}  expected {float(pD[r,68]):.4f}  tile0 key4 {float(pD[r,4]):.4f}  no-m {float(torch.exp(s[r,68])):.4f}"
                      f"  m of row-32 {float(torch.exp(s[r,68]-m0[r-32,0])):.4f}  m0 {float(m0[r,0]):.3f} s {float(s[r,68]):.3f}")
                print(f"      sum w {float(w[r].sum()):.4f}  sum over keys<112: {float(w[r,:112].sum()):.4f} ref {float(wref[r,:112].sum()):.4f}")
