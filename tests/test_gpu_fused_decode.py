"""GPU: the one-launch decode step (nv_attn_decode_paged_fused: q/k-norm + rotary + KV scatter + paged
attention) against the two launches it replaces (nv_qknorm_rope_store, then nv_attn_decode_paged), which the
other parity tests hold to the oracle and to the reference's fixtures.  The fused kernel runs the same
per-head function on the same values and hands the attention the new token's rows as the cache would, so
every cache byte must be equal, and the outputs may differ only where the compiler contracts the fp32
expressions of the two attention instantiations differently (which multiply-add pairs become one fused
operation): at most one bf16 ulp at the scale of the output's row, on at most one output in two hundred (observed:
none on most shapes, 59 of 524,288 on the benchmark's batch).  The slot the new token goes to is filled with NaN beforehand: whatever
the key stream fetched from it must not reach a result."""
import math

import pytest
import torch

from conftest import bits, make_paged_problem

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 128


def _step(seed, ctx, H_q, H_kv, bs, norm, strided=True, max_pos=40960):
    """One decode step: caches holding ctx[b] - 1 tokens per sequence, the raw projections of the new tokens,
    their positions (ctx[b] - 1) and slots."""
    from nanovllm_amd.layers.rope import build_cos_sin_cache
    B = len(ctx)
    nb = sum(math.ceil(c / bs) for c in ctx) + 2
    p = make_paged_problem(seed, [1] * B, ctx, H_q, H_kv, bs, nb, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(seed + 1)
    width = (H_q + 2 * H_kv) * D
    if strided:      # slices of one fused projection, as the model produces them
        qkv = torch.randn(B, width + 64, generator=g, device=DEV).bfloat16()
        q = qkv[:, :H_q * D].unflatten(1, (H_q, D))
        k = qkv[:, H_q * D:(H_q + H_kv) * D].unflatten(1, (H_kv, D))
        v = qkv[:, (H_q + H_kv) * D:width].unflatten(1, (H_kv, D))
    else:
        q = torch.randn(B, H_q, D, generator=g, device=DEV).bfloat16()
        k = torch.randn(B, H_kv, D, generator=g, device=DEV).bfloat16()
        v = torch.randn(B, H_kv, D, generator=g, device=DEV).bfloat16()
    q_w = (1 + 0.1 * torch.randn(D, generator=g, device=DEV)).bfloat16() if norm else None
    k_w = (1 + 0.1 * torch.randn(D, generator=g, device=DEV)).bfloat16() if norm else None
    cos_sin = build_cos_sin_cache(D, max_pos, 1e6).to(DEV).view(max_pos, D)
    pos = torch.tensor([c - 1 for c in ctx], dtype=torch.int64, device=DEV)
    bt = p["block_tables"].cpu()
    slots = torch.tensor([int(bt[b, (c - 1) // bs]) * bs + (c - 1) % bs for b, c in enumerate(ctx)],
                         dtype=torch.int32, device=DEV)
    # the new tokens' slots hold NaN until the step writes them
    for cache in (p["k_cache"], p["v_cache"]):
        cache.view(-1, H_kv, D)[slots.long()] = float("nan")
    return p, q, k, v, q_w, k_w, cos_sin, pos, slots


def _both(ops, p, q, k, v, q_w, k_w, cos_sin, pos, slots, eps=1e-6):
    kc1, vc1 = p["k_cache"].clone(), p["v_cache"].clone()
    q_rot, _ = ops.qknorm_rope_store(pos, q, k, v, q_w, k_w, cos_sin, eps, slots, kc1, vc1, want_k=False)
    o1 = ops.attn_decode_paged(q_rot, kc1, vc1, p["cu_k"], p["block_tables"], p["max_k"], p["scale"])
    kc2, vc2 = p["k_cache"].clone(), p["v_cache"].clone()
    o2 = ops.attn_decode_paged_fused(pos, q, k, v, q_w, k_w, cos_sin, eps, slots, kc2, vc2, p["cu_k"],
                                     p["block_tables"], p["max_k"], p["scale"])
    torch.cuda.synchronize()
    return (o1, kc1, vc1), (o2, kc2, vc2)


def _assert_same(a, b, name):
    for x, y, what in zip(a[1:], b[1:], ("k_cache", "v_cache")):
        assert torch.equal(bits(x), bits(y)), (name, what, "bits differ", int((bits(x) != bits(y)).sum()))
    _assert_out(a[0], b[0], name)


def _assert_out(x, y, name):
    """Equal bits, except where the two forms round differently: which fp32 multiply-add pairs the compiler contracts
    differs between the instantiations, and a launch with few (sequence, kv head) pairs may combine its key ranges
    inside the launch in one form and in the merge launch in the other (another summation order).  Either moves a sum
    by a few fp32 ulps of its TERMS, so an output may differ by one bf16 ulp at the scale of its row (an element far
    smaller than its row's rms carries the row's absolute error), on at most one value in two hundred (observed: none
    on most shapes, up to 1.3 per thousand where the combine forms differ)."""
    assert torch.isfinite(x.float()).all() and torch.isfinite(y.float()).all(), (name, "non-finite output")
    differ = bits(x) != bits(y)
    n = int(differ.sum())
    if n == 0:
        return
    assert n <= max(4, x.numel() // 200), (name, "outputs differ in", n, "of", x.numel())
    xf, yf = x.float(), y.float()
    rms = xf.pow(2).mean(dim=-1, keepdim=True).sqrt().expand_as(xf)
    scale = torch.maximum(torch.maximum(xf.abs(), yf.abs()), rms).clamp_min(2.0 ** -126)
    ulp = torch.exp2(torch.floor(torch.log2(scale)) - 7)
    worst = ((xf - yf).abs() / ulp)[differ].max().item()
    assert worst <= 1.0, (name, "outputs differ by", worst, "bf16 ulp at the row's scale")


@pytest.fixture(scope="module")
def hip():
    from nanovllm_amd import _lib, ops
    _lib.set_tuning()
    yield ops
    _lib.set_tuning()


@pytest.mark.parametrize("norm", [True, False])
@pytest.mark.parametrize("H_q,H_kv", [(16, 8), (8, 8), (2, 1), (4, 4), (32, 8), (64, 8), (8, 1), (24, 8), (40, 8),
                                      (7, 1)])
def test_fused_step_equals_two_launches(hip, H_q, H_kv, norm):
    # ragged contexts: first key, tile and block boundaries on either side, several blocks
    ctx = [1, 2, 15, 16, 17, 255, 256, 257, 300, 511, 512, 513, 700, 768, 1025]
    st = _step(3, ctx, H_q, H_kv, 256, norm)
    a, b = _both(hip, *st)
    _assert_same(a, b, f"H{H_q}/{H_kv} norm={norm}")


@pytest.mark.parametrize("H_q", [16, 32])
@pytest.mark.parametrize("bs", [16, 64, 512])
def test_fused_step_block_sizes(hip, bs, H_q):
    ctx = [1, bs - 1, bs, bs + 1, 3 * bs, 3 * bs + 5, 777]
    st = _step(5, ctx, H_q, 8, bs, True, strided=False)
    a, b = _both(hip, *st)
    _assert_same(a, b, f"H{H_q}/8 block {bs}")


@pytest.mark.parametrize("H_q", [16, 32, 64])
@pytest.mark.parametrize("B,ctx_hi,wi", [(1, 9000, -1), (2, 32768, -1), (3, 5000, 0), (3, 5000, 1), (8, 2100, -1)])
def test_fused_step_with_split_sequences(hip, B, ctx_hi, wi, H_q):
    """Few sequences with long contexts: the keys are cut into work items (waves or workgroups, both
    forced), only the item that holds position n_k - 1 takes the new token, and the merge launch follows."""
    from nanovllm_amd import _lib
    g = torch.Generator().manual_seed(B * 7 + ctx_hi)
    ctx = [int(x) for x in torch.randint(ctx_hi // 2, ctx_hi + 1, (B,), generator=g)]
    ctx[0] = ctx_hi
    if B > 1:
        ctx[1] = ctx_hi // 2 + 1        # a sequence whose last work items are empty
    _lib.set_tuning(decode_work_item=wi)
    try:
        assert _lib.lib().nv_attn_decode_workspace_bytes(B, H_q, 8, D, max(ctx)) > 16, "this case must split"
        st = _step(11, ctx, H_q, 8, 256, True)
        a, b = _both(hip, *st)
        _assert_same(a, b, f"H{H_q}/8 B{B} ctx<={ctx_hi} work item {wi}")
    finally:
        _lib.set_tuning()


def test_fused_step_headline_batch(hip):
    """The benchmark's batch: 256 sequences, contexts U[513, 768], 16 query / 8 kv heads."""
    g = torch.Generator().manual_seed(0)
    ctx = torch.randint(513, 769, (256,), generator=g).tolist()
    st = _step(0, ctx, 16, 8, 256, True)
    a, b = _both(hip, *st)
    _assert_same(a, b, "headline")


@pytest.mark.parametrize("H_q", [16, 32])
def test_fused_step_padded_rows_and_bad_slots(hip, H_q):
    """Rows a captured graph pads its batch with (no keys, slot -1) and a slot outside the cache: nothing is
    written for them, the attention of the out-of-range row still sees its new token."""
    ctx = [300, 40, 513]
    p, q, k, v, q_w, k_w, cos_sin, pos, slots = _step(9, ctx, H_q, 8, 256, True)
    # append two padded rows
    pad = 2
    cu_k = torch.cat([p["cu_k"], p["cu_k"][-1:].repeat(pad)])
    bt = torch.cat([p["block_tables"], torch.full((pad, p["block_tables"].shape[1]), -1, dtype=torch.int32, device=DEV)])
    grow = lambda t: torch.cat([t, torch.zeros((pad,) + t.shape[1:], dtype=t.dtype, device=DEV)])
    q2, k2, v2 = grow(q.contiguous()), grow(k.contiguous()), grow(v.contiguous())
    pos2 = torch.cat([pos, torch.zeros(pad, dtype=torch.int64, device=DEV)])
    slots2 = torch.cat([slots, torch.full((pad,), -1, dtype=torch.int32, device=DEV)])
    slots2[1] = p["k_cache"].shape[0] * 256 + 5          # outside the cache
    p2 = dict(p, cu_k=cu_k, block_tables=bt)
    a, b = _both(hip, p2, q2, k2, v2, q_w, k_w, cos_sin, pos2, slots2)
    # the two-launch path reads row 1's NaN slot (its key was never written); the fused launch has the new
    # token in registers: compare the rows both define, and require the fused row 1 to be finite
    keep = [0, 2]
    _assert_out(a[0][keep], b[0][keep], "rows with a slot")
    assert torch.isfinite(b[0][1].float()).all()
    assert (b[0][3:].float() == 0).all()
    # caches: identical except nothing anywhere for row 1 (both skip the out-of-range slot)
    assert torch.equal(bits(a[1]), bits(b[1])) and torch.equal(bits(a[2]), bits(b[2]))


@pytest.mark.parametrize("H_q", [16, 64])
def test_fused_step_stores_the_token_even_when_max_seqlen_k_is_understated(hip, H_q):
    """A max_seqlen_k below a sequence's key count is a caller's error that costs that sequence its last keys in
    either form; the scatter must not depend on it (a token missing from the cache would spoil every later step)."""
    ctx = [700, 130, 1500]
    p, q, k, v, q_w, k_w, cos_sin, pos, slots = _step(13, ctx, H_q, 8, 256, True)
    p = dict(p, max_k=1024)
    a, b = _both(hip, p, q, k, v, q_w, k_w, cos_sin, pos, slots)
    assert torch.equal(bits(a[1]), bits(b[1])) and torch.equal(bits(a[2]), bits(b[2]))
    _assert_out(a[0][:2], b[0][:2], "rows inside the bound")


def test_fused_step_refuses_what_it_has_no_kernel_for(hip):
    """More than two query heads per kv head run the MFMA-score kernel; with that path switched off by the
    tuning record there is no fused kernel for them, and more than eight heads per kv head have none at all."""
    from nanovllm_amd import _lib
    st = _step(1, [100, 200], 32, 8, 256, True)
    _lib.set_tuning(decode_score_path=0)
    try:
        with pytest.raises(_lib.NanoVllmHipError, match="MFMA score path"):
            _both(hip, *st)
    finally:
        _lib.set_tuning()
    assert hip.decode_fused_supported(32, 8, 256) and hip.decode_fused_supported(16, 8, 256)
    assert not hip.decode_fused_supported(16, 1, 256) and not hip.decode_fused_supported(16, 8, 8)
