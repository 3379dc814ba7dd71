"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container, where the reference checkout is mounted at /root/reference
(it does not exist on the GPU box; the committed .npz/.json files are what travels).

    python tests/golden/make_golden.py

How the reference is imported: its top-level ``nanovllm/__init__.py`` pulls in the engine and,
through ``layers/attention.py``, the third-party ``flash_attn`` package, which is not installed.
The script therefore registers an empty ``nanovllm`` namespace whose ``__path__`` points at the
reference sources (so submodules import normally, only the top-level __init__ is skipped) and a
placeholder ``flash_attn`` module whose only function raises when called.  Nothing that reaches
flash-attn is executed: the attention fixtures (g4) come from the oracle restatement and are
cross-checked here against the reference's dense SDPA helper where that helper is valid.

Fixtures (bf16 tensors are stored as their int16 bit patterns):
  g1_store.npz    Attention.store_kvcache                       (reference output)
  g2_rmsnorm.npz  RMSNorm.forward compiled, with/without residual, hidden and per-head (reference)
  g7_silu.npz     SiluAndMul.forward compiled (reference)
  g3_rope.npz     RotaryPositionEmbedding.forward                (reference output)
  g4_attention.npz  paged varlen attention cases: the reference SDPA helper's output for every
                    full-prefill / decode sequence, the oracle's for chunked ones (marked unpinned)
  g5_hash.json    BlockManager.compute_hash known answers        (reference output)
  g6_paging.json  scheduler / block-manager scenario trace       (reference, can_allocate patched)
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference checkout not present: fixtures can only be regenerated in the build container")
    pkg = types.ModuleType("nanovllm")
    pkg.__path__ = [os.path.join(REF, "nanovllm")]
    sys.modules["nanovllm"] = pkg
    fa = types.ModuleType("flash_attn")

    def _absent(*a, **k):
        raise RuntimeError("flash_attn is not installed; the golden generator must not call it")

    fa.flash_attn_varlen_func = _absent
    sys.modules["flash_attn"] = fa
    from nanovllm.layers.attention import Attention
    from nanovllm.layers.layernorm import RMSNorm
    from nanovllm.layers.rope import RotaryPositionEmbedding
    from nanovllm.engine.block_manager import BlockManager
    from nanovllm.engine.scheduler import Scheduler
    from nanovllm.engine.sequence import Sequence
    from nanovllm.sampling_params import SamplingParams
    return dict(Attention=Attention, RMSNorm=RMSNorm, Rope=RotaryPositionEmbedding, BlockManager=BlockManager,
                Scheduler=Scheduler, Sequence=Sequence, SamplingParams=SamplingParams)


def bits(t: torch.Tensor) -> np.ndarray:
    assert t.dtype == torch.bfloat16
    return t.contiguous().view(torch.int16).numpy().copy()


def g1_store(ref):
    g = torch.Generator().manual_seed(101)
    H_kv, D, NB, bs, T = 8, 128, 4, 256, 7
    k = torch.randn(T, H_kv, D, generator=g).bfloat16()
    v = torch.randn(T, H_kv, D, generator=g).bfloat16()
    # pre-existing cache content: a cheap repeating bit pattern (compresses well, still shows
    # whether any row other than the addressed ones was touched)
    n = NB * bs * H_kv * D
    k_cache = (torch.arange(n, dtype=torch.int32) % 251 + 0x3C00).to(torch.int16).view(torch.bfloat16).view(NB, bs, H_kv, D)
    v_cache = (torch.arange(n, dtype=torch.int32) % 241 + 0x3D00).to(torch.int16).view(torch.bfloat16).view(NB, bs, H_kv, D)
    slots = torch.tensor([254, 255, 256, 257, 3 * 256 + 17, 0, 600], dtype=torch.int32)  # crosses a block edge
    attn = ref["Attention"](16, D, H_kv, D ** -0.5, 4096)
    attn.k_cache = k_cache.clone()
    attn.v_cache = v_cache.clone()
    attn.store_kvcache(k, v, slots, bs)
    # the reference semantics in the small case quoted in the design notes: bs=4, slots 5,6,7,12
    small = ref["Attention"](2, 8, 2, 1.0, 16)
    small.k_cache = torch.zeros(4, 4, 2, 8)
    small.v_cache = torch.zeros(4, 4, 2, 8)
    ks = torch.arange(4 * 2 * 8, dtype=torch.float32).view(4, 2, 8) + 1
    small.store_kvcache(ks, -ks, torch.tensor([5, 6, 7, 12], dtype=torch.int32), 4)
    assert torch.equal(small.k_cache[1, 1], ks[0]) and torch.equal(small.k_cache[1, 3], ks[2])
    assert torch.equal(small.k_cache[3, 0], ks[3]) and torch.equal(small.v_cache[3, 0], -ks[3])
    np.savez_compressed(os.path.join(HERE, "g1_store.npz"), k=bits(k), v=bits(v), k_cache_in=bits(k_cache),
             v_cache_in=bits(v_cache), slot_mapping=slots.numpy(), block_size=np.int32(bs),
             k_cache_out=bits(attn.k_cache), v_cache_out=bits(attn.v_cache))


def g2_rmsnorm(ref):
    from oracle.ref_layers import rmsnorm_ref
    g = torch.Generator().manual_seed(202)
    out = {}
    torch.set_default_dtype(torch.bfloat16)      # the reference builds its model under this default
    try:
        for tag, shape in (("hidden", (64, 1024)), ("head", (8, 16, 128)), ("wide", (5, 5120))):
            hid = shape[-1]
            mod = ref["RMSNorm"](hid, eps=1e-6)
            with torch.no_grad():
                mod.weight.copy_((1.0 + 0.1 * torch.randn(hid, generator=g, dtype=torch.float32)).bfloat16())
            x = (torch.randn(*shape, generator=g, dtype=torch.float32) * 1.7).bfloat16()
            res = torch.randn(*shape, generator=g, dtype=torch.float32).bfloat16()
            with torch.no_grad():
                y = mod(x.clone())                                   # compiled forward (Inductor, CPU)
                y2, new_res = mod(x.clone(), res.clone())
            # the restatement must reproduce the compiled reference exactly
            yo = rmsnorm_ref(x, mod.weight.data, 1e-6)
            yo2, nro = rmsnorm_ref(x, mod.weight.data, 1e-6, res)
            for a, b_, what in ((y, yo, "y"), (y2, yo2, "y_res"), (new_res, nro, "new_res")):
                mism = (a.view(torch.int16) != b_.view(torch.int16)).sum().item()
                print(f"  g2[{tag}] oracle vs compiled reference, {what}: {mism}/{a.numel()} mismatching elements")
            out.update({f"{tag}_x": bits(x), f"{tag}_res": bits(res), f"{tag}_w": bits(mod.weight.data),
                        f"{tag}_y": bits(y), f"{tag}_y_res": bits(y2), f"{tag}_new_res": bits(new_res)})
    finally:
        torch.set_default_dtype(torch.float32)
    out["eps"] = np.float32(1e-6)
    np.savez(os.path.join(HERE, "g2_rmsnorm.npz"), **out)


def g3_rope(ref):
    from oracle.ref_layers import build_cos_sin_cache, rope_ref
    g = torch.Generator().manual_seed(303)
    D, max_pos, base = 128, 40960, 1000000.0
    mod = ref["Rope"](D, D, max_pos, base)
    positions = torch.tensor([0, 1, 2, 3, 17, 255, 256, 511, 512, 1023, 4095, 8191, 16384, 32767, 40959, 40000]
                             + list(range(100, 116)), dtype=torch.int64)
    T = positions.numel()
    q = torch.randn(T, 16, D, generator=g).bfloat16()
    k = torch.randn(T, 8, D, generator=g).bfloat16()
    with torch.no_grad():
        qo, ko = mod(positions, q, k)
    table = build_cos_sin_cache(D, max_pos, base)
    assert torch.equal(table, mod.cos_sin_cache), "oracle cos/sin table differs from the reference's"
    qr, kr = rope_ref(positions, q, k, table)
    print("  g3 oracle vs reference: q mismatches", (qr.view(torch.int16) != qo.view(torch.int16)).sum().item(),
          "k mismatches", (kr.view(torch.int16) != ko.view(torch.int16)).sum().item())
    np.savez(os.path.join(HERE, "g3_rope.npz"), positions=positions.numpy(), q=bits(q), k=bits(k),
             q_out=bits(qo), k_out=bits(ko), cos_sin_rows=mod.cos_sin_cache[positions, 0].numpy(),
             head_dim=np.int32(D), max_pos=np.int32(max_pos), base=np.float64(base))


def _attention_case(g, name, n_qs, n_ks, H_q, H_kv, bs, NB, D=128, coarse=False):
    """Random paged problem: non-monotone physical block ids, -1 padded tables.  coarse=True draws the
    cache from N(0,1) rounded to multiples of 1/8 (few distinct bf16 patterns: the fixture compresses)."""
    B = len(n_qs)
    perm = torch.randperm(NB, generator=g).tolist()
    width = max((n + bs - 1) // bs for n in n_ks) + 1           # at least one padding column
    tables, nxt = [], 0
    for n in n_ks:
        nb = (n + bs - 1) // bs
        tables.append(perm[nxt:nxt + nb] + [-1] * (width - nb))
        nxt += nb
    assert nxt <= NB
    T = sum(n_qs)
    q = torch.randn(T, H_q, D, generator=g).bfloat16()
    k_cache = torch.randn(NB, bs, H_kv, D, generator=g).bfloat16()
    v_cache = torch.randn(NB, bs, H_kv, D, generator=g).bfloat16()
    if coarse:
        k_cache = ((k_cache.float() * 8).round() / 8).bfloat16()
        v_cache = ((v_cache.float() * 8).round() / 8).bfloat16()
    cu_q = torch.tensor([0] + np.cumsum(n_qs).tolist(), dtype=torch.int32)
    cu_k = torch.tensor([0] + np.cumsum(n_ks).tolist(), dtype=torch.int32)
    return dict(name=name, q=q, k_cache=k_cache, v_cache=v_cache, cu_q=cu_q, cu_k=cu_k,
                block_tables=torch.tensor(tables, dtype=torch.int32), scale=D ** -0.5, H_q=H_q, H_kv=H_kv, bs=bs)


def g4_attention(ref):
    """Paged varlen attention cases.  `{name}_ref_pinned[b]` says where the rows of sequence b in
    `{name}_out` come from:
      True  -- from the REFERENCE's own dense helper Attention._scaled_dot_product_attention_batch
               (nanovllm/layers/attention.py:273-335), run here on the gathered keys.  The helper is
               valid for full prefill (n_q == n_k) and decode (n_q == 1) only: its top-left tril
               mask is wrong for chunked prefill.  The oracle must reproduce these rows to 1e-5.
      False -- from the oracle restatement (chunked prefill, 1 < n_q < n_k).  Nothing the reference
               holds can pin these rows: they rest on flash-attn's published bottom-right rule."""
    from oracle.ref_attention import paged_attention_ref, gather_kv
    g = torch.Generator().manual_seed(404)
    cases = [
        _attention_case(g, "full_prefill_g2", [70, 128, 5], [70, 128, 5], 4, 2, 64, 8),
        _attention_case(g, "chunked_prefill_g4", [5, 64], [13, 200], 8, 2, 64, 8),
        _attention_case(g, "chunked_bs256", [37], [300], 4, 2, 256, 3),
        _attention_case(g, "decode_g2", [1, 1, 1, 1], [1, 63, 64, 130], 4, 2, 64, 8),
        _attention_case(g, "decode_g8", [1, 1], [257, 31], 8, 1, 256, 3),
        _attention_case(g, "mixed_g2", [1, 40, 1, 3], [90, 40, 17, 67], 4, 2, 64, 8),
        _attention_case(g, "decode_g1", [1, 1, 1], [16, 33, 7], 2, 2, 64, 4),
        # the reference's block size (256), full prefill and decode, 2 / 4 / 8 query heads per kv head
        _attention_case(g, "bs256_full_prefill_g2", [258, 40], [258, 40], 2, 1, 256, 4, coarse=True),
        _attention_case(g, "bs256_full_prefill_g4", [257, 33], [257, 33], 4, 1, 256, 4, coarse=True),
        _attention_case(g, "bs256_full_prefill_g8", [258], [258], 8, 1, 256, 3, coarse=True),
        _attention_case(g, "bs256_decode_g2", [1, 1, 1], [513, 256, 700], 2, 1, 256, 7, coarse=True),
        _attention_case(g, "bs256_decode_g4", [1, 1, 1], [257, 768, 1], 4, 1, 256, 6, coarse=True),
        _attention_case(g, "bs256_decode_g8", [1, 1], [600, 255], 8, 1, 256, 5, coarse=True),
        _attention_case(g, "bs256_mixed_g4", [1, 200, 1, 70], [300, 200, 40, 326], 4, 1, 256, 7, coarse=True),
    ]
    out = {"names": np.array([c["name"] for c in cases])}
    for c in cases:
        o = paged_attention_ref(c["q"], c["k_cache"], c["v_cache"], c["cu_q"], c["cu_k"], c["block_tables"], c["scale"])
        attn = ref["Attention"](c["H_q"], 128, c["H_kv"], c["scale"], 4096)
        cq, ck = c["cu_q"].tolist(), c["cu_k"].tolist()
        expected = o.clone()
        pinned = np.zeros(len(cq) - 1, dtype=np.bool_)
        for b in range(len(cq) - 1):
            n_q, n_k = cq[b + 1] - cq[b], ck[b + 1] - ck[b]
            if n_q == n_k or n_q == 1:
                kb = gather_kv(c["k_cache"], c["block_tables"][b], n_k).float().unsqueeze(0)
                vb = gather_kv(c["v_cache"], c["block_tables"][b], n_k).float().unsqueeze(0)
                qb = c["q"][cq[b]:cq[b + 1]].float().unsqueeze(0)
                r = attn._scaled_dot_product_attention_batch(qb, kb, vb)[0]
                err = (r - o[cq[b]:cq[b + 1]]).abs().max().item()
                assert err < 1e-5, (c["name"], b, err)
                expected[cq[b]:cq[b + 1]] = r
                pinned[b] = True
        print(f"  g4[{c['name']}]: {int(pinned.sum())}/{len(cq) - 1} sequences carry the reference SDPA helper's output")
        n = c["name"]
        out.update({f"{n}_q": bits(c["q"]), f"{n}_k_cache": bits(c["k_cache"]), f"{n}_v_cache": bits(c["v_cache"]),
                    f"{n}_cu_q": c["cu_q"].numpy(), f"{n}_cu_k": c["cu_k"].numpy(),
                    f"{n}_block_tables": c["block_tables"].numpy(), f"{n}_scale": np.float32(c["scale"]),
                    f"{n}_out": expected.numpy(), f"{n}_ref_pinned": pinned})
    np.savez_compressed(os.path.join(HERE, "g4_attention.npz"), **out)


def g5_hash(ref):
    BM = ref["BlockManager"]
    h1 = BM.compute_hash([1, 2, 3, 4])
    h2 = BM.compute_hash([5, 6, 7, 8], h1)
    long_tokens = list(range(1000, 1256))
    h3 = BM.compute_hash(long_tokens)
    h4 = BM.compute_hash(list(range(256)), h3)
    vectors = [
        dict(tokens=[1, 2, 3, 4], prev=-1, hash=h1),
        dict(tokens=[5, 6, 7, 8], prev=h1, hash=h2),
        dict(tokens=long_tokens, prev=-1, hash=h3),
        dict(tokens=list(range(256)), prev=h3, hash=h4),
    ]
    with open(os.path.join(HERE, "g5_hash.json"), "w") as f:
        json.dump(vectors, f)
    print("  g5:", h1, h2)


def g6_paging(ref):
    """Scripted admit / chunk / prefix-hit / append / free scenario on the reference scheduler.
    The reference's can_allocate calls len() on an int (block_manager.py:317) and cannot run as
    committed; the intended comparison `num_free_blocks >= n` is patched in."""
    import contextlib
    import io
    BM, Scheduler, Sequence, SP = ref["BlockManager"], ref["Scheduler"], ref["Sequence"], ref["SamplingParams"]
    BM.can_allocate = lambda self, n: self.num_free_blocks >= n
    cfg = types.SimpleNamespace(max_num_seqs=4, kvcache_block_size=4, max_num_batched_tokens=4000, eos=-1,
                                num_kvcache_blocks=16)
    sched = Scheduler(cfg)
    sched.long_prefill_token_threshold = 6
    A = Sequence(list(range(10)), SP(temperature=1.0, max_tokens=3))
    Bq = Sequence(list(range(8)) + [99, 98, 97], SP(temperature=1.0, max_tokens=2))
    names = {A.id: "A", Bq.id: "B"}
    sched.add(A)
    steps = []
    next_token = 500
    with contextlib.redirect_stdout(io.StringIO()):
        for step in range(12):
            if step == 2:
                sched.add(Bq)
            if sched.is_finished():
                break
            seqs, _ = sched.schedule()
            steps.append([dict(seq=names[s.id], computed=s.num_computed_tokens, scheduled=s.num_scheduled_tokens,
                               block_table=list(s.block_table)) for s in seqs])
            toks = []
            for s in seqs:
                done = s.num_computed_tokens + s.num_scheduled_tokens >= len(s.prompt_tokens)
                toks.append(next_token if done else None)
                next_token += 1 if done else 0
            sched.update_from_output(seqs, toks)
    trace = dict(block_size=4, num_blocks=16, chunk=6, steps=steps,
                 free_blocks_at_end=sched.block_manager.num_free_blocks,
                 cached_hashes_at_end=len(sched.block_manager.cache_blocks),
                 note="block tables come from the reference as committed: entries appended by "
                      "append_slots appear twice (seq.block_table aliases request_blocks[id])")
    with open(os.path.join(HERE, "g6_paging.json"), "w") as f:
        json.dump(trace, f, indent=1)
    for i, st in enumerate(steps):
        print("  g6 step", i, [(e["seq"], e["computed"], e["scheduled"], e["block_table"]) for e in st])
    print("  g6 end: free", trace["free_blocks_at_end"], "cached", trace["cached_hashes_at_end"])


def g7_silu(ref):
    from nanovllm.layers.activation import SiluAndMul
    from oracle.ref_layers import silu_and_mul_ref
    g = torch.Generator().manual_seed(707)
    mod = SiluAndMul()
    out = {}
    for tag, shape, amp in (("mlp", (48, 2 * 3072), 2.0), ("small", (3, 2 * 64), 1.0), ("large_values", (16, 2 * 512), 12.0)):
        x = (torch.randn(*shape, generator=g, dtype=torch.float32) * amp).bfloat16()
        with torch.no_grad():
            y = mod(x.clone())                                       # compiled forward (Inductor, CPU)
        yo = silu_and_mul_ref(x)
        mism = (y.view(torch.int16) != yo.view(torch.int16)).sum().item()
        print(f"  g7[{tag}] oracle vs compiled reference: {mism}/{y.numel()} mismatching elements")
        out.update({f"{tag}_x": bits(x), f"{tag}_y": bits(y)})
    np.savez(os.path.join(HERE, "g7_silu.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    ref = import_reference()
    every = (g1_store, g3_rope, g4_attention, g5_hash, g6_paging, g2_rmsnorm, g7_silu)
    wanted = sys.argv[1:]                     # e.g. "g7_silu": regenerate only that fixture
    for fn in every:
        if wanted and fn.__name__ not in wanted:
            continue
        print(fn.__name__)
        fn(ref)
    print("done")
