"""GPU: the HIP kernels, called through the C ABI (ctypes), against the CPU oracle and the golden
fixtures.  Bars: bit-exact for the KV scatter and block-table indexing (integer/byte work);
max-abs <= 2e-2 against the fp32 oracle for attention, RMSNorm and rotary outputs (bf16 I/O).
"""
import numpy as np
import pytest
import torch

from conftest import assert_bound_is_sensitive, assert_rows_close, bf16, bits, make_paged_problem
from longctx_util import check_rows_with_sensitivity
from oracle import ref_attention, ref_layers

pytestmark = pytest.mark.gpu

TOL = 2e-2          # max-abs tolerance against the fp32 oracle (inputs ~N(0,1), outputs O(1))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import nanovllm_amd
    from nanovllm_amd import _lib, ops
    _lib.lib()                       # raises if the extension is missing: no fallback
    return nanovllm_amd, ops


@pytest.fixture
def tuning():
    """Force launch-plan choices through nv_set_tuning for one test; heuristics are restored after."""
    from nanovllm_amd import _lib
    yield _lib.set_tuning
    _lib.set_tuning()


def to_dev(p):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in p.items()}


# ------------------------------------------------------------------ store_kvcache (bit exact)
def test_store_matches_reference_fixture(hip, golden):
    _, ops = hip
    g = golden("g1_store.npz")
    k_cache, v_cache = bf16(g["k_cache_in"]).to(DEV), bf16(g["v_cache_in"]).to(DEV)
    ops.store_kvcache(bf16(g["k"]).to(DEV), bf16(g["v"]).to(DEV), k_cache, v_cache,
                      torch.from_numpy(g["slot_mapping"]).to(DEV))
    assert torch.equal(bits(k_cache.cpu()), torch.from_numpy(g["k_cache_out"]))
    assert torch.equal(bits(v_cache.cpu()), torch.from_numpy(g["v_cache_out"]))


@pytest.mark.parametrize("T,H_kv,NB,bs", [(1, 8, 2, 256), (257, 8, 6, 256), (4000, 8, 20, 256), (33, 4, 3, 16), (5, 1, 2, 256)])
def test_store_random_slots_bit_exact(hip, T, H_kv, NB, bs):
    _, ops = hip
    g = torch.Generator().manual_seed(T)
    D = 128
    # k and v are slices of one fused projection output: token stride > row length
    kv = torch.randn(T, 2 * H_kv * D, generator=g).bfloat16()
    k, v = kv[:, :H_kv * D].view(T, H_kv, D), kv[:, H_kv * D:].view(T, H_kv, D)
    slots = torch.randperm(NB * bs, generator=g)[:T].to(torch.int32)
    kc = torch.randn(NB, bs, H_kv, D, generator=g).bfloat16()
    vc = torch.randn(NB, bs, H_kv, D, generator=g).bfloat16()
    kc_ref, vc_ref = kc.clone(), vc.clone()
    ref_attention.store_kvcache_ref(k, v, kc_ref, vc_ref, slots, bs)
    kvd = kv.to(DEV)
    kd, vd = kvd[:, :H_kv * D].view(T, H_kv, D), kvd[:, H_kv * D:].view(T, H_kv, D)
    kcd, vcd = kc.to(DEV), vc.to(DEV)
    ops.store_kvcache(kd, vd, kcd, vcd, slots.to(DEV))
    assert torch.equal(bits(kcd.cpu()), bits(kc_ref)) and torch.equal(bits(vcd.cpu()), bits(vc_ref))


def test_store_skips_negative_and_out_of_range_slots(hip):
    _, ops = hip
    T, H_kv, D, NB, bs = 4, 8, 128, 2, 256
    k = torch.ones(T, H_kv, D, dtype=torch.bfloat16, device=DEV)
    kc = torch.zeros(NB, bs, H_kv, D, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    ops.store_kvcache(k, k, kc, vc, torch.tensor([-1, 5, NB * bs, 7], dtype=torch.int32, device=DEV))
    touched = kc.view(NB * bs, -1).abs().sum(dim=1).nonzero().flatten().tolist()
    assert touched == [5, 7]


def test_store_empty_batch(hip):
    _, ops = hip
    kc = torch.zeros(1, 256, 8, 128, dtype=torch.bfloat16, device=DEV)
    e = torch.zeros(0, 8, 128, dtype=torch.bfloat16, device=DEV)
    ops.store_kvcache(e, e, kc, kc.clone(), torch.zeros(0, dtype=torch.int32, device=DEV))


# ------------------------------------------------------------------ RMSNorm
def _check_norm(got, want_bits_or_tensor, name, max_mismatch=2e-3):
    want = want_bits_or_tensor
    mism = (bits(got) != bits(want)).float().mean().item()
    err = (got.float() - want.float()).abs().max().item()
    assert err <= TOL * max(1.0, want.float().abs().max().item()), (name, err)
    assert mism <= max_mismatch, (name, "fraction of elements differing from the reference by an ulp", mism)


def test_rmsnorm_matches_compiled_reference_fixture(hip, golden):
    _, ops = hip
    g = golden("g2_rmsnorm.npz")
    eps = float(g["eps"])
    for tag in ("hidden", "head", "wide"):
        x, res, w = bf16(g[f"{tag}_x"]).to(DEV), bf16(g[f"{tag}_res"]).to(DEV), bf16(g[f"{tag}_w"]).to(DEV)
        y = ops.rmsnorm(x, w, eps)
        y2, nr = ops.add_rmsnorm(x, res, w, eps)
        assert torch.equal(bits(nr.cpu()), torch.from_numpy(g[f"{tag}_new_res"])), tag     # exact sum
        _check_norm(y.cpu(), bf16(g[f"{tag}_y"]), tag + "_y")
        _check_norm(y2.cpu(), bf16(g[f"{tag}_y_res"]), tag + "_y_res")
        assert y.shape == x.shape and y2.shape == x.shape


@pytest.mark.parametrize("rows,cols", [(1, 128), (3, 1024), (256, 1024), (7, 2048), (5, 4096), (2, 5120), (1, 16384), (300, 128)])
def test_rmsnorm_shapes_against_oracle(hip, rows, cols):
    _, ops = hip
    g = torch.Generator().manual_seed(rows * 131 + cols)
    x = (torch.randn(rows, cols, generator=g) * 2).bfloat16()
    res = torch.randn(rows, cols, generator=g).bfloat16()
    w = (1 + 0.2 * torch.randn(cols, generator=g)).bfloat16()
    y = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6).cpu()
    y2, nr = ops.add_rmsnorm(x.to(DEV), res.to(DEV), w.to(DEV), 1e-6)
    ro = ref_layers.rmsnorm_ref(x, w, 1e-6)
    ro2, rnr = ref_layers.rmsnorm_ref(x, w, 1e-6, res)
    assert torch.equal(bits(nr.cpu()), bits(rnr))
    _check_norm(y, ro, "y")
    _check_norm(y2.cpu(), ro2, "y_res")


def test_rmsnorm_module_and_input_preserved(hip):
    nv, _ = hip
    norm = nv.RMSNorm(1024, eps=1e-6, dtype=torch.bfloat16, device=DEV)
    x = torch.randn(16, 1024, device=DEV).bfloat16()
    res = torch.randn(16, 1024, device=DEV).bfloat16()
    x0, r0 = x.clone(), res.clone()
    y = norm(x)
    y2, nr = norm(x, res)
    assert torch.equal(x, x0) and torch.equal(res, r0)            # inputs untouched
    assert y.dtype == torch.bfloat16 and nr.shape == x.shape
    want, want_nr = ref_layers.rmsnorm_ref(x.cpu(), norm.weight.data.cpu(), 1e-6, res.cpu())
    _check_norm(y2.cpu(), want, "module")
    assert torch.equal(bits(nr.cpu()), bits(want_nr))


# ------------------------------------------------------------------ SiluAndMul
def _check_silu(y, want, name):
    """The kernel's exp differs from the host's in the last fp32 bits: rare flips of one bf16 ulp."""
    mism = bits(y) != bits(want)
    assert mism.float().mean().item() <= 5e-3, (name, mism.float().mean().item())
    yf, wf = y.float(), want.float()
    assert ((yf - wf).abs() <= wf.abs() * 2 ** -7 + 1e-30).all(), name


def test_silu_and_mul_matches_compiled_reference_fixture(hip, golden):
    nv, ops = hip
    from nanovllm_amd.layers.activation import SiluAndMul
    g = golden("g7_silu.npz")
    for tag in ("mlp", "small", "large_values"):
        x = bf16(g[f"{tag}_x"]).to(DEV)
        y = SiluAndMul()(x)
        assert y.shape == (x.shape[0], x.shape[1] // 2)
        _check_silu(y.cpu(), bf16(g[f"{tag}_y"]), tag)


@pytest.mark.parametrize("rows,inter", [(1, 8), (256, 3072), (7, 12288), (300, 1024)])
def test_silu_and_mul_shapes_and_strides_against_oracle(hip, rows, inter):
    _, ops = hip
    g = torch.Generator().manual_seed(rows + inter)
    wide = (torch.randn(rows, 2 * inter + 64, generator=g) * 3).bfloat16()
    wide[0, 0] = -200.0                                            # exp overflow side: result must be -0 * up
    x = wide[:, :2 * inter]                                        # row stride larger than the row
    y = ops.silu_and_mul(x.to(DEV)[:, :]) if rows == 1 else ops.silu_and_mul(wide.to(DEV)[:, :2 * inter])
    _check_silu(y.cpu(), ref_layers.silu_and_mul_ref(x), "strided")
    assert torch.isfinite(y.float()).all()


# ------------------------------------------------------------------ rotary embedding
def test_rope_matches_reference_fixture(hip, golden):
    nv, _ = hip
    g = golden("g3_rope.npz")
    rope = nv.RotaryPositionEmbedding(int(g["head_dim"]), int(g["head_dim"]), int(g["max_pos"]), float(g["base"])).to(DEV)
    qo, ko = rope(torch.from_numpy(g["positions"]).to(DEV), bf16(g["q"]).to(DEV), bf16(g["k"]).to(DEV))
    # separately rounded products and one final rounding: the same fp32 expression tree as the reference
    assert (bits(qo.cpu()) != torch.from_numpy(g["q_out"])).sum().item() == 0
    assert (bits(ko.cpu()) != torch.from_numpy(g["k_out"])).sum().item() == 0


@pytest.mark.parametrize("T,H_q,H_kv", [(1, 16, 8), (256, 16, 8), (37, 64, 8), (5, 32, 8), (3, 4, 4)])
def test_rope_strided_inputs_against_oracle(hip, T, H_q, H_kv):
    _, ops = hip
    D, max_pos = 128, 4096
    g = torch.Generator().manual_seed(T + H_q)
    qkv = torch.randn(T, (H_q + 2 * H_kv) * D, generator=g).bfloat16()
    q = qkv[:, :H_q * D].view(T, H_q, D)
    k = qkv[:, H_q * D:(H_q + H_kv) * D].view(T, H_kv, D)
    pos = torch.randint(0, max_pos, (T,), generator=g, dtype=torch.int64)
    table = ref_layers.build_cos_sin_cache(D, max_pos, 1e6)
    rq, rk = ref_layers.rope_ref(pos, q, k, table)
    qd = qkv.to(DEV)
    qo, ko = ops.rope(pos.to(DEV), qd[:, :H_q * D].view(T, H_q, D), qd[:, H_q * D:(H_q + H_kv) * D].view(T, H_kv, D),
                      table.to(DEV))
    assert torch.equal(bits(qo.cpu()), bits(rq)) and torch.equal(bits(ko.cpu()), bits(rk))


@pytest.mark.parametrize("norm", [True, False])
@pytest.mark.parametrize("T,H_q,H_kv", [(1, 16, 8), (256, 16, 8), (19, 64, 8), (512, 32, 8)])
def test_fused_qknorm_rope_store_against_oracle(hip, T, H_q, H_kv, norm):
    _, ops = hip
    D, max_pos, NB, bs = 128, 2048, 4, 256
    g = torch.Generator().manual_seed(T * 7 + H_q)
    q = torch.randn(T, H_q, D, generator=g).bfloat16()
    k = torch.randn(T, H_kv, D, generator=g).bfloat16()
    v = torch.randn(T, H_kv, D, generator=g).bfloat16()
    qw = (1 + 0.1 * torch.randn(D, generator=g)).bfloat16() if norm else None
    kw = (1 + 0.1 * torch.randn(D, generator=g)).bfloat16() if norm else None
    pos = torch.randint(0, max_pos, (T,), generator=g, dtype=torch.int64)
    slots = torch.randperm(NB * bs, generator=g)[:T].to(torch.int32)
    table = ref_layers.build_cos_sin_cache(D, max_pos, 1e6)
    kc = torch.randn(NB, bs, H_kv, D, generator=g).bfloat16()
    vc = torch.randn(NB, bs, H_kv, D, generator=g).bfloat16()
    rq, rk = ref_layers.qknorm_rope_ref(pos, q, k, qw, kw, table, 1e-6)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    ref_attention.store_kvcache_ref(rk, v, kc_ref, vc_ref, slots, bs)
    kcd, vcd = kc.to(DEV), vc.to(DEV)
    qo, ko = ops.qknorm_rope_store(pos.to(DEV), q.to(DEV), k.to(DEV), v.to(DEV),
                                   qw.to(DEV) if norm else None, kw.to(DEV) if norm else None,
                                   table.to(DEV), 1e-6, slots.to(DEV), kcd, vcd)
    if norm:
        _check_norm(qo.cpu(), rq, "q", max_mismatch=5e-3)
        _check_norm(ko.cpu(), rk, "k", max_mismatch=5e-3)
    else:
        assert torch.equal(bits(qo.cpu()), bits(rq)) and torch.equal(bits(ko.cpu()), bits(rk))
    # the scatter itself is exact: the cache holds exactly what the kernel returned as k, and v
    kc_exp, vc_exp = kc.clone(), vc.clone()
    ref_attention.store_kvcache_ref(ko.cpu(), v, kc_exp, vc_exp, slots, bs)
    assert torch.equal(bits(kcd.cpu()), bits(kc_exp)) and torch.equal(bits(vcd.cpu()), bits(vc_ref))


# ------------------------------------------------------------------ attention
def _attn_check(o, ref, name):
    err = (o.float().cpu() - ref).abs().max().item()
    assert err <= TOL, (name, err)
    return err


def _run_attention(ops, p, mode):
    if mode == "decode":
        return ops.attn_decode_paged(p["q"], p["k_cache"], p["v_cache"], p["cu_k"], p["block_tables"], p["max_k"],
                                     p["scale"], cu_seqlens_q=p["cu_q"])
    if mode == "prefill":
        return ops.attn_prefill_varlen(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"], p["block_tables"],
                                       p["max_q"], p["max_k"], p["scale"])
    o = ops.attn_prefill_varlen(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"], p["block_tables"],
                                p["max_q"], p["max_k"], p["scale"], skip_single_query=True)
    return ops.attn_decode_paged(p["q"], p["k_cache"], p["v_cache"], p["cu_k"], p["block_tables"], p["max_k"],
                                 p["scale"], cu_seqlens_q=p["cu_q"], out=o)


def test_attention_golden_cases(hip, golden):
    """HIP against the stored expectations of g4 directly: for full-prefill and decode sequences those
    are the outputs of the reference's own dense SDPA helper (nanovllm/layers/attention.py:273-335),
    for chunked-prefill sequences the oracle's (unpinned by the reference; see the fixture's mask)."""
    _, ops = hip
    g = golden("g4_attention.npz")
    for name in g["names"]:
        cu_q, cu_k = g[f"{name}_cu_q"], g[f"{name}_cu_k"]
        n_q = np.diff(cu_q)
        p = dict(q=bf16(g[f"{name}_q"]), k_cache=bf16(g[f"{name}_k_cache"]), v_cache=bf16(g[f"{name}_v_cache"]),
                 cu_q=torch.from_numpy(cu_q), cu_k=torch.from_numpy(cu_k),
                 block_tables=torch.from_numpy(g[f"{name}_block_tables"]), scale=float(g[f"{name}_scale"]),
                 max_q=int(n_q.max()), max_k=int(np.diff(cu_k).max()))
        p = to_dev(p)
        ref = torch.from_numpy(g[f"{name}_out"])
        modes = ["mixed", "prefill"] + (["decode"] if n_q.max() == 1 else [])
        for mode in modes:
            _attn_check(_run_attention(ops, p, mode), ref, f"{name}/{mode}")


@pytest.mark.parametrize("kernel", [0, 1])
def test_chunked_prefill_reproduces_reference_pinned_full_prefill_rows(hip, golden, tuning, kernel):
    """The chunk-invariance anchor of tests/test_oracle_golden.py on the GPU, both prefill kernels: a chunk
    of rows lo..hi-1 over hi cached keys must reproduce those rows of the full-prefill output, and the
    stored full-prefill outputs are the reference's own dense helper's (nanovllm/layers/attention.py:273-335)."""
    from test_oracle_golden import FULL_PREFILL_CASES, chunk_problems
    _, ops = hip
    g = golden("g4_attention.npz")
    tuning(prefill_kernel=kernel)
    chunks = 0
    for name in FULL_PREFILL_CASES:
        want = torch.from_numpy(g[f"{name}_out"])
        kc, vc = bf16(g[f"{name}_k_cache"]).to(DEV), bf16(g[f"{name}_v_cache"]).to(DEV)
        for p, rows in chunk_problems(g, name):
            pd = to_dev(dict(p, k_cache=kc, v_cache=vc, scale=float(g[f"{name}_scale"])))
            _attn_check(_run_attention(ops, pd, "prefill"), want[rows], f"{name} rows {rows} kernel {kernel}")
            chunks += 1
    assert chunks >= 20


@pytest.mark.parametrize("H_q,H_kv", [(16, 8), (32, 8), (64, 8), (8, 8), (8, 1)])
@pytest.mark.parametrize("bs", [256, 16])
def test_decode_ragged_against_oracle(hip, H_q, H_kv, bs):
    _, ops = hip
    n_ks = [1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 513, 700, 768]
    nb = sum((n + bs - 1) // bs for n in n_ks) + 2
    p = make_paged_problem(H_q * 3 + bs, [1] * len(n_ks), n_ks, H_q, H_kv, bs, nb)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    pd = to_dev(p)
    o = ops.attn_decode_paged(pd["q"], pd["k_cache"], pd["v_cache"], pd["cu_k"], pd["block_tables"], p["max_k"],
                              p["scale"])                                   # cu_seqlens_q omitted: token b = seq b
    _attn_check(o, ref, "decode")
    o2 = _run_attention(ops, pd, "decode")
    assert torch.equal(o, o2)


def _row_samples(p, rows_per_seq, heads):
    """(token row, head, block-table row, visible keys) for chosen query rows of each sequence."""
    cq, ck = p["cu_q"].tolist(), p["cu_k"].tolist()
    out = []
    for b, rows in enumerate(rows_per_seq):
        n_q, n_k = cq[b + 1] - cq[b], ck[b + 1] - ck[b]
        for r in rows:
            for h in heads:
                out.append((cq[b] + r, h, p["block_tables"][b], n_k - n_q + r + 1))
    return out


@pytest.mark.parametrize("peaked", [False, True])
@pytest.mark.parametrize("single", [0, 1])
def test_decode_long_context_split(hip, tuning, single, peaked):
    """Few sequences, long contexts: many key chunks per (sequence, head), combined by the merge launch
    (single = 0) or inside the launch (single = 1).  Every (sequence, head) row under the scale-aware bound,
    whose sensitivity is shown per row (the oracle without one 256-key block violates it); ``peaked``: softmax
    scale x 4, a handful of keys carry the weight."""
    _, ops = hip
    n_ks = [4097, 3000, 8192]
    p = make_paged_problem(5, [1, 1, 1], n_ks, 32, 8, 256, 64)
    if peaked:
        p["scale"] *= 4.0
    pd = to_dev(p)
    tuning(decode_single_launch=single)
    o = _run_attention(ops, pd, "decode").cpu()
    worst, least = check_rows_with_sensitivity(lambda t, h: o[t, h], _row_samples(p, [[0]] * 3, range(32)), p["q"],
                                               p["k_cache"], p["v_cache"], 4, p["scale"], "decode-long", peaked,
                                               assert_rows_close, assert_bound_is_sensitive)
    print(f"decode-long single={single} peaked={peaked}: worst err/bound {worst:.3f}, least damage/bound {least:.1f}")


@pytest.mark.parametrize("H_q,H_kv", [(16, 8), (32, 8), (64, 8)])
@pytest.mark.parametrize("direction", ["rising", "falling"])
def test_decode_monotone_scores(hip, H_q, H_kv, direction):
    """Scores that grow (or shrink) steadily along the context: the running maximum is raised many
    times (or exactly once), which exercises the accumulator rescale of the online softmax."""
    _, ops = hip
    n_ks = [300, 1000, 2049]
    bs = 256
    p = make_paged_problem(77, [1] * len(n_ks), n_ks, H_q, H_kv, bs, 16)
    g = torch.Generator().manual_seed(5)
    u = torch.randn(128, generator=g)
    p["q"] = (u[None, None, :] + 0.05 * torch.randn(len(n_ks), H_q, 128, generator=g)).bfloat16()
    for b, n in enumerate(n_ks):
        for j0 in range(0, n, bs):
            blk = p["block_tables"][b, j0 // bs].item()
            j = torch.arange(j0, min(j0 + bs, n), dtype=torch.float32)
            ramp = (j / n) if direction == "rising" else (1.0 - j / n)
            rows = 6.0 * ramp[:, None, None] * u[None, None, :] + 0.1 * torch.randn(len(j), H_kv, 128, generator=g)
            p["k_cache"][blk, : len(j)] = rows.bfloat16()
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    pd = to_dev(p)
    _attn_check(_run_attention(ops, pd, "decode"), ref, f"decode-{direction}")


@pytest.mark.parametrize("single", [0, 1])
@pytest.mark.parametrize("form", ["0", "1"])
@pytest.mark.parametrize("H_q,H_kv,bs", [(16, 8, 256), (32, 8, 256), (8, 1, 256), (4, 4, 48)])
def test_decode_both_work_item_forms(hip, tuning, form, single, H_q, H_kv, bs):
    """Work item = wave (0) or workgroup whose four waves split the chunk and merge in LDS (1), each
    forced, with and without a global split (long and short contexts, empty quarters, 48-token blocks);
    the partial states of split sequences combined by a second launch (single = 0) or inside the launch
    by the work item that arrives last at the pair's counter (single = 1).  The in-launch form runs three
    times in a row on one workspace: every launch has to leave the counters at zero, and the results must
    be bit-identical from launch to launch and to the two-launch form's (same partial states, and both
    combines must agree to fp32 rounding; the comparison is against the oracle at the usual tolerance)."""
    _, ops = hip
    n_ks = [1, 17, 63, 64, 65, 255, 700, 3000, 5000]
    nb = sum((n + bs - 1) // bs for n in n_ks) + 2
    p = make_paged_problem(H_q + bs, [1] * len(n_ks), n_ks, H_q, H_kv, bs, nb)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    pd = to_dev(p)
    tuning(decode_work_item=int(form), decode_single_launch=single, decode_chunk=256)
    outs = [_run_attention(ops, pd, "decode") for _ in range(3 if single else 1)]
    for o in outs:
        _attn_check(o, ref, f"decode-form{form}-single{single}")
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "in-launch combine differs between consecutive launches"
    tuning(decode_work_item=int(form), decode_single_launch=single)
    _attn_check(_run_attention(ops, pd, "decode"), ref, f"decode-form{form}-single{single}-auto-chunk")


@pytest.mark.parametrize("H_q,H_kv", [(12, 4), (40, 8), (24, 4), (28, 4)])
def test_attention_odd_group_sizes(hip, H_q, H_kv):
    """3, 5, 6 and 7 query heads per kv head (e.g. 40 / 8): decode runs the kernel laid out for the
    next of 4 / 8 heads with zero-query surplus columns, prefill shares K/V tiles between gcd(G, 4) heads."""
    _, ops = hip
    n_qs, n_ks = [1, 1, 70, 1, 129], [1, 300, 70, 2049, 400]
    p = make_paged_problem(H_q, n_qs, n_ks, H_q, H_kv, 256, 16)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    pd = to_dev(p)
    _attn_check(_run_attention(ops, pd, "mixed"), ref, "mixed")
    _attn_check(_run_attention(ops, pd, "prefill"), ref, "prefill")
    dec = make_paged_problem(H_q + 1, [1] * 4, [5, 64, 1000, 5000], H_q, H_kv, 256, 28)
    ref = ref_attention.paged_attention_ref(dec["q"], dec["k_cache"], dec["v_cache"], dec["cu_q"], dec["cu_k"],
                                            dec["block_tables"], dec["scale"])
    _attn_check(_run_attention(ops, to_dev(dec), "decode"), ref, "decode")


def test_decode_zero_length_sequence_gives_zeros(hip):
    _, ops = hip
    p = make_paged_problem(9, [1, 1], [0, 40], 16, 8, 256, 3)
    p["block_tables"][0, :] = -1
    pd = to_dev(p)
    o = _run_attention(ops, pd, "decode").cpu().float()
    assert o[0].abs().max().item() == 0.0
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    _attn_check(o, ref, "decode-empty")


def test_decode_ignores_uninitialised_cache_rows(hip):
    """Rows of the cache beyond a sequence's length may hold NaN bit patterns (torch.empty)."""
    _, ops = hip
    p = make_paged_problem(11, [1, 1, 1], [5, 70, 257], 16, 8, 256, 5)
    nan = torch.tensor(float("nan")).bfloat16()
    for b, n in enumerate([5, 70, 257]):
        for j in range(n, ((n + 255) // 256) * 256):
            blk = p["block_tables"][b, j // 256].item()
            p["k_cache"][blk, j % 256] = nan
            p["v_cache"][blk, j % 256] = nan
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    pd = to_dev(p)
    _attn_check(_run_attention(ops, pd, "decode"), ref, "decode-nan-tail")
    _attn_check(_run_attention(ops, pd, "prefill"), ref, "prefill-nan-tail")


@pytest.mark.parametrize("H_q,H_kv", [(16, 8), (32, 8), (4, 4)])
@pytest.mark.parametrize("n_qs,n_ks", [
    ([512], [512]),                         # full prefill, one sequence
    ([1, 31, 32, 33, 127, 128, 129], [1, 31, 32, 33, 127, 128, 129]),   # ragged full prefill
    ([512, 512, 300], [1024, 512, 812]),    # chunked prefill: keys already cached
    ([5], [13]),
    ([64, 1, 200, 1], [64, 300, 456, 1]),   # mixed with decode rows
])
def test_prefill_against_oracle(hip, H_q, H_kv, n_qs, n_ks):
    _, ops = hip
    bs = 256
    nb = sum((n + bs - 1) // bs for n in n_ks) + 1
    p = make_paged_problem(sum(n_ks) + H_q, n_qs, n_ks, H_q, H_kv, bs, nb)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    pd = to_dev(p)
    _attn_check(_run_attention(ops, pd, "prefill"), ref, "prefill")
    _attn_check(_run_attention(ops, pd, "mixed"), ref, "mixed")


@pytest.mark.parametrize("H_q,H_kv", [(4, 1), (16, 8), (32, 8), (4, 4), (64, 8), (12, 4)])
@pytest.mark.parametrize("n_qs,n_ks", [
    ([64], [64]), ([64], [192]), ([256], [256]), ([70], [70]), ([5], [13]), ([3], [1]),
    ([1, 31, 32, 33, 127, 128, 129], [1, 31, 32, 33, 127, 128, 129]),
    ([512, 512, 300], [1024, 512, 812]),
    ([64, 1, 200, 1], [64, 300, 456, 1]),
    ([300], [2049]),
])
def test_prefill_hand_scheduled_kernel_against_oracle(hip, tuning, H_q, H_kv, n_qs, n_ks):
    """The hand-scheduled kernel (one wave per SIMD, 64 rows per wave, software-pipelined asm loop), forced
    for launches of any size: one tile, several tiles, masked diagonal and tail tiles, chunked prefill,
    rows without any visible key, decode rows inside the launch, every head-group shape."""
    _, ops = hip
    bs = 256
    nb = sum((n + bs - 1) // bs for n in n_ks) + 1
    p = make_paged_problem(sum(n_ks) + 3 * H_q, n_qs, n_ks, H_q, H_kv, bs, nb)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    pd = to_dev(p)
    tuning(prefill_kernel=1)
    _attn_check(_run_attention(ops, pd, "prefill"), ref, "prefill-hand")
    _attn_check(_run_attention(ops, pd, "mixed"), ref, "mixed-hand")


def test_prefill_hand_scheduled_kernel_rescale_and_nan_tails(hip, tuning):
    """Side paths of the hand-scheduled loop: a late key with a much larger score (the running maximum
    jumps: accumulator rescale), scores that rise steadily along the context (many rescales), NaN bit
    patterns in the cache rows behind the end of a sequence (the tail tile re-reads its last valid row),
    block size 64 and a strided q."""
    _, ops = hip
    tuning(prefill_kernel=1)
    p = make_paged_problem(3, [512], [512], 4, 2, 256, 3)
    blk = p["block_tables"][0, 1].item()
    p["k_cache"][blk, 200] = (p["q"][511, 0] * 3).to(torch.bfloat16)[None, :].expand(2, -1)   # spike for query 511
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    _attn_check(_run_attention(ops, to_dev(p), "prefill"), ref, "hand-spike")
    p = make_paged_problem(8, [700], [700], 8, 2, 256, 4)
    g = torch.Generator().manual_seed(5)
    u = torch.randn(128, generator=g)
    p["q"] = (u[None, None, :] + 0.05 * torch.randn(700, 8, 128, generator=g)).bfloat16()
    for j0 in range(0, 700, 256):
        blk = p["block_tables"][0, j0 // 256].item()
        j = torch.arange(j0, min(j0 + 256, 700), dtype=torch.float32)
        rows = 6.0 * (j / 700)[:, None, None] * u[None, None, :] + 0.1 * torch.randn(len(j), 2, 128, generator=g)
        p["k_cache"][blk, : len(j)] = rows.bfloat16()
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    _attn_check(_run_attention(ops, to_dev(p), "prefill"), ref, "hand-rising")
    p = make_paged_problem(11, [5, 70, 257], [5, 70, 257], 16, 8, 256, 5)
    nan = torch.tensor(float("nan")).bfloat16()
    for b, n in enumerate([5, 70, 257]):
        for j in range(n, ((n + 255) // 256) * 256):
            blk = p["block_tables"][b, j // 256].item()
            p["k_cache"][blk, j % 256] = nan
            p["v_cache"][blk, j % 256] = nan
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    _attn_check(_run_attention(ops, to_dev(p), "prefill"), ref, "hand-nan-tail")
    p = make_paged_problem(77, [100, 260], [164, 260], 8, 2, 64, 9)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    pd = to_dev(p)
    T, H_q, D = p["q"].shape
    wide = torch.zeros(T, H_q * D + 512, dtype=torch.bfloat16, device=DEV)
    wide[:, :H_q * D] = pd["q"].view(T, -1)
    pd["q"] = wide[:, :H_q * D].view(T, H_q, D)
    _attn_check(_run_attention(ops, pd, "prefill"), ref, "hand-bs64-strided")


@pytest.mark.parametrize("n_qs,n_ks", [
    ([1024] * 16, [1024] * 16),                                   # 4 query tiles per head: pairs (3, 0), (2, 1)
    ([768] * 22, [768] * 22),                                     # 3 tiles: the middle one is a lone second item
    ([960, 1000, 64, 70] * 4, [960, 1300, 64, 326] * 4),          # ragged and chunked sequences in one launch
    ([1024] * 32, [1024] * 32),                                   # four tiles per workgroup: 16 tiles per head
    ([1000, 330, 64, 700] * 8, [1300, 330, 326, 700] * 8),        # four per workgroup, ragged: 16, 6, 1 and 11 tiles
])
def test_prefill_hand_scheduled_kernel_two_tiles_per_workgroup(hip, tuning, n_qs, n_ks):
    """Launches with many workgroups give every workgroup two or four query tiles of one head (chosen from
    both ends of the sequence so that all workgroups have equal work): the next tile's first key tiles and
    its Q fragments are fetched while the current tile finishes.  A key with a much larger score late in the visible range of a
    row of the FIRST query tile (processed second) drives the accumulator rescale through that path too."""
    _, ops = hip
    H_q, H_kv, bs = 16, 2, 256
    nb = sum((n + bs - 1) // bs for n in n_ks) + 1
    p = make_paged_problem(len(n_qs) + n_qs[0], n_qs, n_ks, H_q, H_kv, bs, nb)
    shift = n_ks[0] - n_qs[0]
    j = shift + 90                                                # visible to query rows >= 90 of sequence 0
    blk = p["block_tables"][0, j // bs].item()
    p["k_cache"][blk, j % bs] = (p["q"][100, 3] * 3).to(torch.bfloat16)[None, :].expand(H_kv, -1)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    tuning(prefill_kernel=1)
    _attn_check(_run_attention(ops, to_dev(p), "prefill"), ref, "prefill-hand-pairs")


@pytest.mark.parametrize("seed", range(12))
def test_prefill_hand_scheduled_kernel_random_ragged_batches(hip, tuning, seed):
    """Seeded random batches for the hand-scheduled kernel: 1 - 6 sequences, query chunks of 1 - 700 rows over
    0 - 1500 cached keys, head shapes with 1, 2, 4 and 8 query heads per kv head, block size 64 or 256, and a
    forced key-range split in every third case (empty ranges, ranges that end inside a tile)."""
    _, ops = hip
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    H_kv = [1, 2, 4, 8][ri(0, 3)]
    H_q = H_kv * [1, 2, 4, 8][ri(0, 3)]
    bs = [64, 256][ri(0, 1)]
    B = ri(1, 6)
    n_qs = [ri(1, 700) if ri(0, 4) else ri(1, 3) for _ in range(B)]
    n_ks = [nq + (ri(0, 1500) if ri(0, 2) else 0) for nq in n_qs]
    nb = sum((n + bs - 1) // bs for n in n_ks) + 2
    p = make_paged_problem(seed + 500, n_qs, n_ks, H_q, H_kv, bs, nb)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    tuning(prefill_kernel=1, prefill_split=[0, 0, 3][seed % 3])
    _attn_check(_run_attention(ops, to_dev(p), "prefill"), ref, f"hand-random-{seed}: H {H_q}/{H_kv} bs {bs} {n_qs} {n_ks}")


def test_prefill_hand_scheduled_kernel_on_engine_step_shapes_with_guard_rows(hip):
    """The launches of an engine prefill step (16 / 8 heads, 8 - 10 ragged sequences, chunk tails of a few
    rows, q a strided slice of a fused qkv buffer, query offsets that do not start at 0), 100 launch pairs:
    the hand-scheduled kernel must agree with the compiler-scheduled one and must not touch the canary
    rows in front of and behind the output (a wave whose rows lie behind the end of its sequence stores
    nothing)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import prefill_stress
    launches, bad = prefill_stress.run(rounds=10, verbose=False)
    assert launches == 100 and bad == 0


def test_prefill_key_range_split_on_chunk_shapes_with_guard_rows_and_poisoned_workspace(hip):
    """The chunks of one long prompt (32 / 8 heads, 64 - 512 queries over 1024 - 8192 cached keys: four key ranges,
    bf16 states through the workspace, merge launch), 50 launch pairs: the hand-scheduled kernel must agree with the
    compiler-scheduled one, must not touch the canary rows around the output, and must not read a state that this
    launch did not write (the workspace holds NaN bit patterns in front of every launch)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import prefill_stress
    launches, bad = prefill_stress.run(rounds=5, H_q=32, H_kv=8, verbose=False, shapes=prefill_stress.SPLIT_SHAPES,
                                       poison_workspace=True)
    assert launches == 50 and bad == 0


@pytest.mark.parametrize("kernel", [-1, 0, 1])
def test_prefill_key_range_split_long_chunk(hip, tuning, kernel):
    """One long sequence prefilled in a chunk: too few workgroups for the chip, so the launch splits
    the key range over several workgroups and merges their states in a second launch (either kernel,
    and whichever the launch plan picks)."""
    _, ops = hip
    from nanovllm_amd import _lib
    H_q, H_kv, n_q, n_k = 32, 8, 512, 4096
    tuning(prefill_kernel=kernel)
    assert _lib.lib().nv_attn_prefill_workspace_bytes(1, n_q, n_q, n_k, H_q, H_kv, 128) > 16   # the split path is taken
    p = make_paged_problem(21, [n_q], [n_k], H_q, H_kv, 256, 17)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    _attn_check(_run_attention(ops, to_dev(p), "prefill"), ref, "prefill-split")


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("nsplit,H_q,H_kv", [(2, 16, 8), (3, 16, 8), (7, 16, 8), (3, 4, 4), (2, 16, 4), (3, 16, 2), (2, 12, 4)])
def test_prefill_forced_key_range_split_ragged(hip, tuning, nsplit, H_q, H_kv, kernel):
    """Forced split counts on a ragged mixed batch, both kernels, head groups of 1, 2, 3, 4 and 8 query heads per
    kv head (a workgroup of the hand-scheduled kernel then holds 4, 2 or 1 row blocks of 64 per head, and its
    merge kernel must find the same rows): splits that lie wholly above a row block's diagonal are empty,
    sequences with one query stay with the decode kernel."""
    _, ops = hip
    n_qs, n_ks = [1, 70, 300, 5, 129], [40, 70, 700, 261, 1000]
    p = make_paged_problem(31, n_qs, n_ks, H_q, H_kv, 256, 12)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    pd = to_dev(p)
    tuning(prefill_split=nsplit, prefill_kernel=kernel)
    _attn_check(_run_attention(ops, pd, "prefill"), ref, f"prefill-split{nsplit}")
    _attn_check(_run_attention(ops, pd, "mixed"), ref, f"mixed-split{nsplit}")


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("nsplit", [0, 3])
@pytest.mark.parametrize("H_q,H_kv,n_seqs", [(8, 1, 1), (16, 2, 1), (8, 1, 3), (32, 4, 1), (12, 6, 1), (32, 4, 3), (16, 4, 5),
                                             (8, 1, 7), (64, 8, 1)])
def test_prefill_pair_counts_that_do_not_fill_the_xcds(hip, tuning, H_q, H_kv, n_seqs, nsplit, kernel):
    """1, 2, 3, 4, 6, 12 and 20 (sequence, kv head) pairs: each pair's workgroups are spread over 8, 4 or 2 XCDs
    (head shards of a large model with one or a few sequences in the step); 7 and 8 pairs keep one XCD per pair.
    Ragged chunked sequences, both kernels, with and without a forced key-range split (whose merge kernel repeats
    the workgroup decomposition)."""
    _, ops = hip
    n_qs = [300, 129, 64, 5, 200, 70, 257][:n_seqs]
    n_ks = [1100, 129, 700, 261, 200, 900, 300][:n_seqs]
    p = make_paged_problem(41 + n_seqs, n_qs, n_ks, H_q, H_kv, 256, 20)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    tuning(prefill_split=nsplit, prefill_kernel=kernel)
    _attn_check(_run_attention(ops, to_dev(p), "prefill"), ref, f"pairs {n_seqs * H_kv} split {nsplit} kernel {kernel}")


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("scale", [0.0, -0.0884, 1.0, 0.003, 0.25, 0.0884 * 1.9999])
def test_prefill_softmax_scale_values(hip, tuning, kernel, scale):
    """The hand-scheduled kernel splits scale * log2(e) into a power of two (folded into Q, exact) and a factor in
    [1, 2) applied in fp32: zero (uniform attention over the visible keys), negative, exact powers of two, a factor
    just below the next power, very peaked and very flat softmax -- chunked, ragged and masked tiles, both kernels."""
    _, ops = hip
    n_qs, n_ks = [200, 64, 5, 300], [456, 64, 261, 1100]
    p = make_paged_problem(41, n_qs, n_ks, 8, 2, 256, 12)
    p["scale"] = scale
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], scale)
    tuning(prefill_kernel=kernel)
    o = _run_attention(ops, to_dev(p), "prefill")
    _attn_check(o, ref, f"scale {scale} kernel {kernel}")
    assert_rows_close(o, ref, f"scale {scale} kernel {kernel}", c=0.04)


@pytest.mark.parametrize("scale", [0.0, -0.0884, 1.0, 0.003])
def test_decode_softmax_scale_values(hip, scale):
    """Decode with zero, negative, large and small softmax scales, short contexts and split ones."""
    _, ops = hip
    n_ks = [1, 65, 700, 5000]
    p = make_paged_problem(43, [1] * 4, n_ks, 16, 8, 256, 30)
    p["scale"] = scale
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], scale)
    o = _run_attention(ops, to_dev(p), "decode")
    _attn_check(o, ref, f"decode scale {scale}")
    assert_rows_close(o, ref, f"decode scale {scale}", c=0.04)


def test_prefill_block_size_64_and_strided_q(hip):
    _, ops = hip
    p = make_paged_problem(77, [100, 260], [164, 260], 8, 2, 64, 9)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    pd = to_dev(p)
    T, H_q, D = p["q"].shape
    wide = torch.zeros(T, H_q * D + 512, dtype=torch.bfloat16, device=DEV)
    wide[:, :H_q * D] = pd["q"].view(T, -1)
    pd["q"] = wide[:, :H_q * D].view(T, H_q, D)                 # token stride larger than the row
    _attn_check(_run_attention(ops, pd, "prefill"), ref, "prefill-bs64")


def test_prefill_online_softmax_rescale_is_exercised(hip):
    """A late key with a much larger score forces the running max to jump in the last tile."""
    _, ops = hip
    p = make_paged_problem(3, [256], [256], 4, 2, 256, 2)
    blk = p["block_tables"][0, 0].item()
    p["k_cache"][blk, 250] = (p["q"][255, 0] * 3).to(torch.bfloat16)[None, :].expand(2, -1)   # spike for query 255
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    _attn_check(_run_attention(ops, to_dev(p), "prefill"), ref, "prefill-spike")
    p2 = make_paged_problem(4, [1], [600], 4, 2, 256, 4)
    blk = p2["block_tables"][0, 2].item()
    p2["k_cache"][blk, 80] = (p2["q"][0, 1] * 3).to(torch.bfloat16)[None, :].expand(2, -1)
    ref2 = ref_attention.paged_attention_ref(p2["q"], p2["k_cache"], p2["v_cache"], p2["cu_q"], p2["cu_k"],
                                             p2["block_tables"], p2["scale"])
    _attn_check(_run_attention(ops, to_dev(p2), "decode"), ref2, "decode-spike")


def test_attention_module_forward_stores_then_attends(hip):
    """Attention.forward: scatter the step's K/V, then attend through the cache (mixed batch)."""
    nv, _ = hip
    H_q, H_kv, D, bs, NB = 16, 8, 128, 256, 6
    n_qs, n_ks = [1, 130, 1], [300, 130, 17]
    p = make_paged_problem(21, n_qs, n_ks, H_q, H_kv, bs, NB)
    T = sum(n_qs)
    g = torch.Generator().manual_seed(5)
    k_new = torch.randn(T, H_kv, D, generator=g).bfloat16()
    v_new = torch.randn(T, H_kv, D, generator=g).bfloat16()
    slots = []
    for b, (nq, nk) in enumerate(zip(n_qs, n_ks)):
        for pos in range(nk - nq, nk):
            slots.append(p["block_tables"][b, pos // bs].item() * bs + pos % bs)
    slots = torch.tensor(slots, dtype=torch.int32)
    kc_ref, vc_ref = p["k_cache"].clone(), p["v_cache"].clone()
    ref = ref_attention.attention_forward_ref(p["q"], k_new, v_new, kc_ref, vc_ref, slots, p["cu_q"], p["cu_k"],
                                              p["block_tables"], p["scale"], bs)
    attn = nv.Attention(H_q, D, H_kv, p["scale"], 4096)
    attn.k_cache, attn.v_cache = p["k_cache"].to(DEV), p["v_cache"].to(DEV)
    # hints a caller may give: none; the number of one-token sequences; or "the first `lead` sequences are
    # decode rows" with per-kernel bounds, which launches each kernel over its own rows only
    lead_hint = dict(num_leading_decode=1, max_seqlen_k_decode=300, max_seqlen_q_rest=130, max_seqlen_k_rest=130)
    for hints in (dict(), dict(num_decode_seqs=2), lead_hint, dict(num_leading_decode=0, max_seqlen_q_rest=130,
                                                                  max_seqlen_k_rest=300)):
        nv.set_context(nv.CacheContext(slot_mapping=slots.to(DEV), cu_seqlens_q=p["cu_q"].to(DEV),
                                       cu_seqlens_k=p["cu_k"].to(DEV), max_seqlen_q=max(n_qs), max_seqlen_k=max(n_ks),
                                       block_tables=p["block_tables"].to(DEV), block_size=bs, **hints))
        o = attn(p["q"].to(DEV), k_new.to(DEV), v_new.to(DEV))
        _attn_check(o, ref, f"module {hints}")
    assert torch.equal(bits(attn.k_cache.cpu()), bits(kc_ref)) and torch.equal(bits(attn.v_cache.cpu()), bits(vc_ref))
    nv.set_context(None)


@pytest.mark.parametrize("peaked", [False, True])
def test_full_size_decode_properties(hip, peaked):
    """BASELINE size (Qwen3-0.6B shapes, B=256, ctx 513..768, block 256): checked through
    size-independent properties instead of the slow oracle: (1) V == const -> output == const;
    (2) softmax-weighted outputs are convex combinations: |o| <= max|v|; (3) a sample of
    sequences x all heads against the oracle under the scale-aware bound, with the bound's sensitivity
    shown per row, also with a peaked softmax (scale x 4); (4) permuting the physical blocks (and the
    tables with them) leaves the output bit-identical."""
    _, ops = hip
    B, H_q, H_kv, D, bs = 256, 16, 8, 128, 256
    g = torch.Generator().manual_seed(0)
    ctx = torch.randint(513, 769, (B,), generator=g).tolist()
    NB = sum((c + bs - 1) // bs for c in ctx)
    p = make_paged_problem(0, [1] * B, ctx, H_q, H_kv, bs, NB, pad_cols=0)
    if peaked:
        p["scale"] *= 4.0
    pd = to_dev(p)
    o = _run_attention(ops, pd, "decode")
    assert o.float().abs().max().item() <= p["v_cache"].float().abs().max().item() + 1e-3
    oc = o.cpu()
    sample = [0, 17, 101, 128, 200, 255]
    rows = [[0] if b in sample else [] for b in range(B)]
    worst, least = check_rows_with_sensitivity(lambda t, h: oc[t, h], _row_samples(p, rows, range(H_q)), p["q"],
                                               p["k_cache"], p["v_cache"], 2, p["scale"], "full-size decode", peaked,
                                               assert_rows_close, assert_bound_is_sensitive)
    print(f"full-size decode peaked={peaked}: worst err/bound {worst:.3f}, least damage/bound {least:.1f}")
    if peaked:
        return
    const = pd["v_cache"].clone().fill_(0.75)
    oc = ops.attn_decode_paged(pd["q"], pd["k_cache"], const, pd["cu_k"], pd["block_tables"], p["max_k"], p["scale"])
    assert (oc.float() - 0.75).abs().max().item() <= 4e-3
    perm = torch.randperm(NB, generator=g)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(NB)
    k2, v2 = p["k_cache"][perm].to(DEV), p["v_cache"][perm].to(DEV)      # new block i holds old block perm[i]
    bt2 = inv[p["block_tables"].long()].to(torch.int32).to(DEV)
    o2 = ops.attn_decode_paged(pd["q"], k2, v2, pd["cu_k"], bt2, p["max_k"], p["scale"])
    assert torch.equal(o, o2)


def test_decode_over_150k_keys(hip):
    """Two sequences x 150,001 and 3 keys (587 blocks; a context longer than any benchmark's, next to a tiny one):
    many key chunks per (sequence, kv head) and a wide block table.  Rows under the scale-aware bound."""
    _, ops = hip
    n_ks, bs = [150001, 3], 256
    p = make_paged_problem(17, [1, 1], n_ks, 8, 2, bs, sum((n + bs - 1) // bs for n in n_ks) + 1)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    o = _run_attention(ops, to_dev(p), "decode")
    _attn_check(o, ref, "150k keys")
    assert_rows_close(o, ref, "150k keys")


@pytest.mark.parametrize("kernel", [0, 1])
def test_prefill_chunk_over_more_keys_than_the_hand_scheduled_kernels_tile_table_holds(hip, tuning, kernel):
    """A 96-token chunk over 70,000 cached keys: more 64-key tiles than the hand-scheduled kernel's table of tile
    addresses holds (1024), so the launch plan has to take the other kernel even when the hand-scheduled one is
    forced; key-range split and merge on a context longer than any benchmark's.  Rows under the scale-aware bound."""
    _, ops = hip
    n_q, n_k, bs = 96, 70000, 256
    p = make_paged_problem(13, [n_q], [n_k], 8, 2, bs, (n_k + bs - 1) // bs + 1)
    ref = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                            p["block_tables"], p["scale"])
    tuning(prefill_kernel=kernel)
    o = _run_attention(ops, to_dev(p), "prefill")
    _attn_check(o, ref, f"70k keys kernel {kernel}")
    assert_rows_close(o, ref, f"70k keys kernel {kernel}")


@pytest.mark.parametrize("path", ["decode", "decode_g8", "prefill0", "prefill1", "chunk0", "chunk1"])
def test_full_size_power_of_two_equivariance(hip, tuning, path):
    """Size-independent and BIT-EXACT at the benchmark sizes: powers of two commute with every rounding on the
    path (bf16 inputs, fp32 products and sums, bf16 states and outputs), so (1) V x 2 doubles the output exactly,
    (2) q x 2 with scale / 2 and (3) K x 4 with scale / 4 leave every score, hence every output bit, unchanged.
    Any step whose result depends on absolute magnitudes breaks one of them: a threshold or clamp kept in the wrong
    units (the running-maximum thresholds are relative to 1 / scale), an intermediate with a fixed exponent range,
    a flush of small values.  Decode at config 2 (B 256, ctx 513..768) and at
    a G = 8 shard shape, whole-prompt prefill of 8192 tokens and its last 512-token chunk (key-range split), both
    prefill kernels."""
    _, ops = hip
    bs = 256
    if path.startswith("decode"):
        H_q, H_kv = (16, 8) if path == "decode" else (16, 2)
        g = torch.Generator().manual_seed(3)
        B = 256 if path == "decode" else 64
        ctx = torch.randint(513, 769, (B,), generator=g).tolist() if path == "decode" else [2048] * B
        p = make_paged_problem(3, [1] * B, ctx, H_q, H_kv, bs, sum((c + bs - 1) // bs for c in ctx), pad_cols=0)
        mode = "decode"
    else:
        H_q, H_kv, n = 32, 8, 8192
        p = make_paged_problem(6, [n], [n], H_q, H_kv, bs, n // bs, pad_cols=0)
        tuning(prefill_kernel=int(path[-1]))
        mode = "prefill"
    pd = to_dev(p)
    if path.startswith("chunk"):
        pd = dict(pd, q=pd["q"][n - 512:].contiguous(), cu_q=torch.tensor([0, 512], dtype=torch.int32, device=DEV), max_q=512)
    o = _run_attention(ops, pd, mode)
    assert torch.isfinite(o.float()).all()
    o_v2 = _run_attention(ops, dict(pd, v_cache=pd["v_cache"] * 2), mode)
    assert torch.equal(o_v2, (o.float() * 2).bfloat16()), "V x 2 must double the output exactly"
    o_q2 = _run_attention(ops, dict(pd, q=pd["q"] * 2, scale=p["scale"] / 2), mode)
    assert torch.equal(o_q2, o), "q x 2 with scale / 2 must not change a bit"
    o_k4 = _run_attention(ops, dict(pd, k_cache=pd["k_cache"] * 4, scale=p["scale"] / 4), mode)
    assert torch.equal(o_k4, o), "K x 4 with scale / 4 must not change a bit"


@pytest.mark.parametrize("peaked", [False, True])
@pytest.mark.parametrize("kernel", [0, 1])
def test_full_size_prefill_properties(hip, tuning, kernel, peaked):
    """Config-3 size (H 32/8, one 8192-token sequence, block 256) through size-independent properties:
    (1) the last 512-token chunk computed alone (keys 0..8191 cached; the key-range split path) equals
    rows 7680..8191 of the whole-prompt prefill (the unsplit path) under the scale-aware row bound; (2) V ==
    const gives const; (3) a sample of rows x 8 heads against the oracle under that bound, whose sensitivity
    is shown per row (the oracle without one 256-key block violates it), from the whole-prompt launch and
    from the chunk launch; (4) permuting the physical blocks leaves the output bit-identical.  ``peaked``:
    softmax scale x 4 (a few keys carry the weight, outputs O(1)): any rounding of scaled scores or scaled
    queries below fp32 shows here first."""
    _, ops = hip
    H_q, H_kv, n, bs, chunk = 32, 8, 8192, 256, 512
    NB = n // bs
    p = make_paged_problem(5, [n], [n], H_q, H_kv, bs, NB, pad_cols=0)
    if peaked:
        p["scale"] *= 4.0
    pd = to_dev(p)
    tuning(prefill_kernel=kernel)        # 0: compiler-scheduled kernel (and its key-range split); 1: hand-scheduled
    full = _run_attention(ops, pd, "prefill")
    last = dict(pd, q=pd["q"][n - chunk:].contiguous(), cu_q=torch.tensor([0, chunk], dtype=torch.int32, device=DEV),
                max_q=chunk)
    part = _run_attention(ops, last, "prefill")
    # two roundings of the same fp32 answer: twice the row bound between them
    assert_rows_close(part, full[n - chunk:], f"chunk vs whole prompt, kernel {kernel}", c=0.06)
    fc, pc = full.cpu(), part.cpu()
    heads = [0, 5, 10, 15, 16, 21, 26, 31]
    rows = [0, 1, 63, 64, 600, 2047, 4100, 7679, 7680, 7999, 8191]
    samples = _row_samples(p, [rows], heads)
    worst, least = check_rows_with_sensitivity(lambda t, h: fc[t, h], samples, p["q"], p["k_cache"], p["v_cache"], 4,
                                               p["scale"], f"full-size prefill kernel {kernel}", peaked,
                                               assert_rows_close, assert_bound_is_sensitive)
    w2, l2 = check_rows_with_sensitivity(lambda t, h: pc[t - (n - chunk), h], [s_ for s_ in samples if s_[0] >= n - chunk],
                                         p["q"], p["k_cache"], p["v_cache"], 4, p["scale"],
                                         f"full-size last chunk kernel {kernel}", peaked, assert_rows_close,
                                         assert_bound_is_sensitive)
    print(f"full-size prefill kernel={kernel} peaked={peaked}: worst err/bound {max(worst, w2):.3f}, "
          f"least damage/bound {min(least, l2):.1f}")
    if peaked:
        return
    const = pd["v_cache"].clone().fill_(-1.5)
    oc = ops.attn_prefill_varlen(pd["q"], pd["k_cache"], const, pd["cu_q"], pd["cu_k"], pd["block_tables"], n, n, p["scale"])
    assert (oc.float() + 1.5).abs().max().item() <= 8e-3
    g = torch.Generator().manual_seed(9)
    perm = torch.randperm(NB, generator=g)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(NB)
    k2, v2 = p["k_cache"][perm].to(DEV), p["v_cache"][perm].to(DEV)
    bt2 = inv[p["block_tables"].long()].to(torch.int32).to(DEV)
    o2 = ops.attn_prefill_varlen(pd["q"], k2, v2, pd["cu_q"], pd["cu_k"], bt2, n, n, p["scale"])
    assert torch.equal(full, o2)


@pytest.mark.parametrize("kernel", [0, 1])
def test_full_size_shared_prefix_prefill_then_decode(hip, tuning, kernel):
    """Config 3's prefix sharing on the kernels at size (H 32/8, block 256): sequence A prefills 8192 tokens;
    sequence B, in the SAME launch, has 256 new queries over 8192 keys and its first 31 block-table entries
    ARE A's physical blocks, which is what the prefix-cache lookup hands a second request
    (nanovllm/engine/block_manager.py:92-127).  B's last block is a private copy of A's last block and B's
    queries are A's last 256, so B's rows must equal A's rows 7936..8191 (two roundings of one fp32 answer);
    sampled rows of both against the oracle under the scale-aware bound with its sensitivity shown.  Then one
    decode step of both sequences (one query each over 8192 keys, 31 shared blocks)."""
    _, ops = hip
    H_q, H_kv, n, bs, new = 32, 8, 8192, 256, 256
    per = n // bs
    p = make_paged_problem(17, [n, new], [n, bs], H_q, H_kv, bs, per + 1, pad_cols=0)   # B gets one private block
    bt = p["block_tables"].clone()
    b_own = bt[1, 0].item()
    bt[1, :per - 1] = bt[0, :per - 1]                    # the shared prefix: A's physical blocks
    bt[1, per - 1] = b_own
    p["k_cache"][b_own] = p["k_cache"][bt[0, per - 1]]
    p["v_cache"][b_own] = p["v_cache"][bt[0, per - 1]]
    p["q"][n:] = p["q"][n - new:n]
    p.update(block_tables=bt, cu_k=torch.tensor([0, n, 2 * n], dtype=torch.int32), max_k=n)
    pd = to_dev(p)
    tuning(prefill_kernel=kernel)
    o = _run_attention(ops, pd, "prefill")
    assert_rows_close(o[n:], o[n - new:n], f"B's rows vs A's last {new} rows, kernel {kernel}", c=0.06)
    oc = o.cpu()
    heads = [0, 7, 12, 18, 25, 31]
    samples = _row_samples(p, [[0, 255, 256, 4095, 7935, 7936, 8000, 8191], [0, 1, 100, 255]], heads)
    worst, least = check_rows_with_sensitivity(lambda t, h: oc[t, h], samples, p["q"], p["k_cache"], p["v_cache"], 4,
                                               p["scale"], f"shared-prefix prefill kernel {kernel}", False,
                                               assert_rows_close, assert_bound_is_sensitive)
    # the decode step that follows: one query per sequence, the same tables
    g = torch.Generator().manual_seed(3)
    dq = torch.randn(2, H_q, 128, generator=g).bfloat16()
    dec = dict(p, q=dq, cu_q=torch.tensor([0, 1, 2], dtype=torch.int32), max_q=1)
    od = _run_attention(ops, to_dev(dec), "decode").cpu()
    w2, l2 = check_rows_with_sensitivity(lambda t, h: od[t, h], _row_samples(dec, [[0], [0]], range(H_q)), dq,
                                         p["k_cache"], p["v_cache"], 4, p["scale"], "shared-prefix decode", False,
                                         assert_rows_close, assert_bound_is_sensitive)
    print(f"shared prefix kernel={kernel}: worst err/bound {max(worst, w2):.3f}, least damage/bound {min(least, l2):.1f}")


def test_unsupported_arguments_raise(hip):
    """Shapes the kernels are not built for are refused with the library's error text, never run."""
    _, ops = hip
    from nanovllm_amd._lib import NanoVllmHipError
    p = to_dev(make_paged_problem(1, [1, 1], [10, 20], 16, 1, 256, 3))          # 16 query heads per kv head
    with pytest.raises(NanoVllmHipError, match="H_q/H_kv"):
        _run_attention(ops, p, "decode")
    p = to_dev(make_paged_problem(1, [40], [40], 8, 4, 32, 3))                  # prefill needs block_size % 64 == 0
    with pytest.raises(NanoVllmHipError, match="block_size"):
        _run_attention(ops, p, "prefill")
    p = to_dev(make_paged_problem(1, [1], [40], 8, 4, 256, 3, D=64))            # head_dim 64
    with pytest.raises((NanoVllmHipError, ValueError)):
        _run_attention(ops, p, "decode")
    p = to_dev(make_paged_problem(1, [1], [600], 8, 4, 256, 4))
    p["block_tables"] = p["block_tables"][:, :2].contiguous()                   # table too narrow for 600 keys
    with pytest.raises(NanoVllmHipError, match="block table"):
        _run_attention(ops, p, "decode")
