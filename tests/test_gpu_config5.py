"""GPU: BASELINE config 5 (Llama-3.1-8B shapes: 32 query / 8 kv heads x 128, no q/k-norm, llama3-scaled
rotary table, contexts of 32768 tokens = 128 cache blocks per sequence) through the C ABI.

The reference cannot run this model (nanovllm/layers/rope.py:67 asserts ``rope_scaling is None``), so the
oracle's llama3 rule is its own restatement of the published formula (oracle/ref_layers.py, pinned on the
CPU in tests/test_oracle_golden.py).  Bars as everywhere: rotary outputs and the KV scatter bit-exact,
attention max-abs <= 2e-2 against the fp32 oracle.
"""
import pytest
import torch

from conftest import assert_bound_is_sensitive, assert_rows_close, bits
from longctx_util import check_rows_with_sensitivity
from oracle import ref_attention, ref_layers, ref_modules

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-2
LLAMA3 = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
              original_max_position_embeddings=8192)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from nanovllm_amd import _lib, ops
    _lib.lib()
    return ops


@pytest.fixture(scope="module")
def long_cache():
    """8 sequences x 32768 keys, H_kv 8, block 256: 1024 shuffled physical blocks (537 MB each for K, V)."""
    B, n_k, H_kv, D, bs = 8, 32768, 8, 128, 256
    per = n_k // bs
    NB = B * per
    gen = torch.Generator(device=DEV).manual_seed(55)
    k = torch.randn(NB, bs, H_kv, D, generator=gen, device=DEV, dtype=torch.float32).bfloat16()
    v = torch.randn(NB, bs, H_kv, D, generator=gen, device=DEV, dtype=torch.float32).bfloat16()
    g = torch.Generator().manual_seed(56)
    bt = torch.randperm(NB, generator=g).view(B, per).to(torch.int32)
    q = torch.randn(B, 32, D, generator=g).bfloat16()
    return dict(k=k, v=v, bt=bt, q=q, n_k=n_k, bs=bs, per=per, NB=NB)


def _samples(c, B, lens, G=4):
    """Every sequence x 4 query heads, walking over all 8 kv heads as b grows."""
    return [(b, (b + 9 * i) % 32, c["bt"][b], lens[b]) for b in range(B) for i in range(4)]


@pytest.mark.parametrize("peaked", [False, True])
@pytest.mark.parametrize("B", [1, 4, 8])
def test_config5_decode_32k_context(ops, long_cache, B, peaked):
    """ctx 32768, H 32/8, B sequences per replica: every sequence x 4 heads against the oracle under the
    scale-aware bound (a 32 k-key average of N(0,1) values has rms 0.009: the fixed 2e-2 is two sigma of
    the ANSWER there and cannot fail), with the bound's sensitivity shown on every sample (the oracle's
    own answer without one 256-key block violates it); ``peaked`` multiplies the softmax scale by 4 so
    that a handful of keys anywhere in the context carry the weight and outputs are O(1).  V == const
    gives const, and a permutation of the physical blocks leaves the output bit-identical.
    B = 1 and B = 4 cut every (sequence, kv head) into 32 key chunks (workgroup and wave work items
    respectively) and combine them with the 8-group merge; B = 8 uses 16 chunks."""
    from nanovllm_amd import _lib
    c = long_cache
    n_k, per = c["n_k"], c["per"]
    ns = (_lib.lib().nv_attn_decode_workspace_bytes(B, 32, 8, 128, n_k) - 65536) // (B * 32 * 132 * 4)
    assert ns == (16 if B == 8 else 32), ns
    scale = 128 ** -0.5 * (4.0 if peaked else 1.0)
    lens = [n_k - 255 * i for i in range(B)]                       # ragged tails inside the last blocks
    cu_k = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32, device=DEV)
    bt = c["bt"][:B].to(DEV)
    q = c["q"][:B].to(DEV)
    o = ops.attn_decode_paged(q, c["k"], c["v"], cu_k, bt, n_k, scale)
    oc = o.cpu()
    worst, least = check_rows_with_sensitivity(lambda t, h: oc[t, h], _samples(c, B, lens), c["q"], c["k"], c["v"], 4,
                                               scale, f"config5 B={B} peaked={peaked}", peaked,
                                               assert_rows_close, assert_bound_is_sensitive)
    print(f"config5 B={B} peaked={peaked}: worst err/bound {worst:.3f}, least damage/bound {least:.1f}")
    if peaked:
        return
    const = torch.full_like(c["v"], 0.625)
    oc = ops.attn_decode_paged(q, c["k"], const, cu_k, bt, n_k, scale)
    del const
    assert (oc.float() - 0.625).abs().max().item() <= 4e-3
    g = torch.Generator().manual_seed(B)
    perm = torch.randperm(c["NB"], generator=g)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(c["NB"])
    k2 = c["k"][perm.to(DEV)]
    v2 = c["v"][perm.to(DEV)]
    bt2 = inv[c["bt"][:B].long()].to(torch.int32).to(DEV)
    o2 = ops.attn_decode_paged(q, k2, v2, cu_k, bt2, n_k, scale)
    assert torch.equal(o, o2)


@pytest.mark.parametrize("single", [0, 1])
@pytest.mark.parametrize("B", [1, 4, 8])
def test_config5_decode_32k_spike_key_in_every_key_chunk(ops, long_cache, B, single):
    """Every key chunk of every (sequence, kv head) must reach the combine: one key whose score exceeds all
    others by > 25 (its K row = 3 q of the kv head's first query head) is planted in chunk
    (rot + b * 8 + hkv) % 32 for rot = 0..31, so that each pair sees it in each chunk once; the output of that
    query head then IS the key's V row (the rest of the softmax weighs < 1e-9), bit for bit.  A chunk that
    is dropped, combined with a wrong weight or read from a wrong block shows as a mismatch.  Both combine
    forms: second launch (single = 0) and inside the launch (single = 1)."""
    from nanovllm_amd import _lib
    c = long_cache
    n_k, bs, G, H_kv = c["n_k"], c["bs"], 4, 8
    lens = [n_k - 255 * i for i in range(B)]
    cu_k = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32, device=DEV)
    bt = c["bt"][:B].to(DEV)
    q = c["q"][:B].to(DEV)
    spike = (q[:, ::G].float() * 3).bfloat16()                     # [B, H_kv, D]
    nchunk = 32
    lens_d = torch.tensor(lens, device=DEV)
    bsel = torch.arange(B, device=DEV)[:, None].expand(B, H_kv)
    hsel = torch.arange(H_kv, device=DEV)[None, :].expand(B, H_kv)
    _lib.set_tuning(decode_single_launch=single)
    try:
        for rot in range(nchunk):
            chunk = (rot + bsel * H_kv + hsel) % nchunk
            pos = torch.minimum(chunk * (n_k // nchunk) + 517, lens_d[:, None] - 2)     # inside the sequence
            blk = bt.long().gather(1, pos // bs)                               # [B, H_kv] physical blocks
            off = pos % bs
            saved = c["k"][blk, off, hsel].clone()
            c["k"][blk, off, hsel] = spike
            try:
                o = ops.attn_decode_paged(q, c["k"], c["v"], cu_k, bt, n_k, 128 ** -0.5)
                want = c["v"][blk, off, hsel]                                  # [B, H_kv, D]
                got = o[:, ::G]
                bad = (bits(got) != bits(want)).any(dim=-1).nonzero().tolist()
                assert not bad, (f"B={B} single={single} rot={rot}: (sequence, kv head) pairs whose output is not the "
                                 f"spike key's V row", bad[:8])
            finally:
                c["k"][blk, off, hsel] = saved
    finally:
        _lib.set_tuning()


def test_llama3_rotary_bit_exact_up_to_position_131071(ops):
    """nv_rope and the fused kernel without q/k-norm weights on the llama3-scaled table: bit-exact
    against the oracle at positions across the whole 131072-entry table."""
    D, max_pos, T, H_q, H_kv = 128, 131072, 96, 32, 8
    table = ref_layers.build_cos_sin_cache(D, max_pos, 5e5, LLAMA3)
    g = torch.Generator().manual_seed(31)
    pos = torch.cat([torch.tensor([0, 1, 8191, 8192, 32767, 32768, 65535, 131070, 131071]),
                     torch.randint(0, max_pos, (T - 9,), generator=g)]).to(torch.int64)
    q = torch.randn(T, H_q, D, generator=g).bfloat16()
    k = torch.randn(T, H_kv, D, generator=g).bfloat16()
    v = torch.randn(T, H_kv, D, generator=g).bfloat16()
    rq, rk = ref_layers.rope_ref(pos, q, k, table)
    td = table.to(DEV)
    qo, ko = ops.rope(pos.to(DEV), q.to(DEV), k.to(DEV), td)
    assert torch.equal(bits(qo.cpu()), bits(rq)) and torch.equal(bits(ko.cpu()), bits(rk))
    NB, bs = 2, 256
    slots = torch.randperm(NB * bs, generator=g)[:T].to(torch.int32)
    kc = torch.zeros(NB, bs, H_kv, D, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    ref_attention.store_kvcache_ref(rk, v, kc_ref, vc_ref, slots, bs)
    kcd, vcd = kc.to(DEV), vc.to(DEV)
    qo2, ko2 = ops.qknorm_rope_store(pos.to(DEV), q.to(DEV), k.to(DEV), v.to(DEV), None, None, td, 1e-5,
                                     slots.to(DEV), kcd, vcd)
    assert torch.equal(bits(qo2.cpu()), bits(rq)) and torch.equal(bits(ko2.cpu()), bits(rk))
    assert torch.equal(bits(kcd.cpu()), bits(kc_ref)) and torch.equal(bits(vcd.cpu()), bits(vc_ref))


def test_llama_shaped_engine_greedy_parity():
    """Llama-3.1-style model on the path (no q/k-norm, llama3-scaled rotary table, 4 query heads per kv
    head), one prompt longer than the 8192 original positions: the HIP engine runs free, the engine built
    from the oracle's modules is teacher-forced on its tokens; logits agree within 2e-2 at every step
    and a token may differ only inside a numerical tie."""
    from test_gpu_engine import _drive, build
    from nanovllm_amd.engine.llm_engine import LLMEngine
    from nanovllm_amd.model.qwen3 import ModelConfig
    cfg = ModelConfig(vocab_size=4096, hidden_size=512, intermediate_size=1024, num_hidden_layers=3,
                      num_attention_heads=16, num_key_value_heads=4, head_dim=128, max_position_embeddings=16384,
                      rms_norm_eps=1e-5, rope_theta=5e5, qk_norm=False, rope_scaling=dict(LLAMA3))
    g = torch.Generator().manual_seed(2)
    prompts = [torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist() for n in (16, 8300, 502)]
    hip_model = build(cfg, None, True, seed=7)
    ref_model = build(cfg, ref_modules, False, seed=7)
    ref_model.load_state_dict(hip_model.state_dict())
    assert not hasattr(hip_model.model.layers[0].self_attn, "q_norm")
    kw = dict(block_size=256, num_kvcache_blocks=44, max_num_batched_tokens=1024, long_prefill_token_threshold=512,
              device=DEV, eos=-1)
    steps = 48
    hip_trace = _drive(LLMEngine(hip_model, cfg, **kw), prompts, steps)
    hip_tokens = [[t for t, _ in tr] for tr in hip_trace]
    ref_trace = _drive(LLMEngine(ref_model, cfg, **kw), prompts, steps, forced=hip_tokens)
    assert all(len(t) == steps for t in hip_tokens)
    ties = 0
    for tr_h, tr_r in zip(hip_trace, ref_trace):
        for (tok, lg_h), (_, lg_r) in zip(tr_h, tr_r):
            scale = max(1.0, lg_r.abs().max().item())
            assert (lg_h - lg_r).abs().max().item() <= 2e-2 * scale
            if tok != int(lg_r.argmax()):
                ties += 1
                assert (lg_r.max() - lg_r[tok]).item() <= 2e-2 * scale, "token differs outside a numerical tie"
    assert ties <= 4, f"{ties} of {3 * steps} greedy tokens fell on numerical ties"
