"""CPU: the oracle restatement against the golden fixtures captured from the reference."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN, bf16, bits
from oracle import ref_attention, ref_layers, ref_metadata


def test_store_kvcache_matches_reference_fixture(golden):
    g = golden("g1_store.npz")
    k_cache, v_cache = bf16(g["k_cache_in"]).clone(), bf16(g["v_cache_in"]).clone()
    ref_attention.store_kvcache_ref(bf16(g["k"]), bf16(g["v"]), k_cache, v_cache,
                                    torch.from_numpy(g["slot_mapping"]), int(g["block_size"]))
    assert torch.equal(bits(k_cache), torch.from_numpy(g["k_cache_out"]))
    assert torch.equal(bits(v_cache), torch.from_numpy(g["v_cache_out"]))


def test_rmsnorm_matches_compiled_reference(golden):
    g = golden("g2_rmsnorm.npz")
    eps = float(g["eps"])
    for tag in ("hidden", "head", "wide"):
        x, res, w = bf16(g[f"{tag}_x"]), bf16(g[f"{tag}_res"]), bf16(g[f"{tag}_w"])
        y = ref_layers.rmsnorm_ref(x, w, eps)
        y2, nr = ref_layers.rmsnorm_ref(x, w, eps, res)
        assert torch.equal(bits(nr), torch.from_numpy(g[f"{tag}_new_res"]))          # bit exact
        for got, key in ((y, f"{tag}_y"), (y2, f"{tag}_y_res")):
            want = bf16(g[key])
            mism = (bits(got) != bits(want)).float().mean().item()
            assert mism <= 1e-3, (tag, key, mism)           # reduction order: rare 1-ulp flips
            assert (got.float() - want.float()).abs().max().item() <= 2e-2 * max(1.0, want.float().abs().max().item())


def test_silu_and_mul_matches_compiled_reference(golden):
    g = golden("g7_silu.npz")
    for tag in ("mlp", "small", "large_values"):
        y = ref_layers.silu_and_mul_ref(bf16(g[f"{tag}_x"]))
        assert torch.equal(bits(y), torch.from_numpy(g[f"{tag}_y"])), tag          # bit exact


def test_rmsnorm_eager_form_is_within_one_ulp(golden):
    g = golden("g2_rmsnorm.npz")
    x, w = bf16(g["hidden_x"]), bf16(g["hidden_w"])
    a = ref_layers.rmsnorm_ref(x, w, 1e-6).float()
    b = ref_layers.rmsnorm_ref_eager(x, w, 1e-6).float()
    assert ((a - b).abs() <= a.abs() * 2 ** -7 + 1e-6).all()


def test_rope_matches_reference_fixture(golden):
    g = golden("g3_rope.npz")
    D, max_pos, base = int(g["head_dim"]), int(g["max_pos"]), float(g["base"])
    table = ref_layers.build_cos_sin_cache(D, max_pos, base)
    pos = torch.from_numpy(g["positions"])
    assert torch.equal(table[pos, 0], torch.from_numpy(g["cos_sin_rows"]))
    qo, ko = ref_layers.rope_ref(pos, bf16(g["q"]), bf16(g["k"]), table)
    assert torch.equal(bits(qo), torch.from_numpy(g["q_out"]))
    assert torch.equal(bits(ko), torch.from_numpy(g["k_out"]))


def test_attention_oracle_equals_stored_reference_outputs(golden):
    """g4: rows of sequences marked `ref_pinned` were produced by the reference's own dense helper
    (nanovllm/layers/attention.py:273-335) when the fixture was generated; the oracle must equal
    them to 1e-5.  Every full-prefill and decode sequence is pinned that way; chunked-prefill
    sequences (1 < n_q < n_k) are not pinnable by anything in the reference and only checked for
    reproducibility."""
    g = golden("g4_attention.npz")
    pinned_seqs = unpinned_seqs = 0
    for name in g["names"]:
        cu_q, cu_k = g[f"{name}_cu_q"], g[f"{name}_cu_k"]
        o = ref_attention.paged_attention_ref(bf16(g[f"{name}_q"]), bf16(g[f"{name}_k_cache"]),
                                              bf16(g[f"{name}_v_cache"]), torch.from_numpy(cu_q), torch.from_numpy(cu_k),
                                              torch.from_numpy(g[f"{name}_block_tables"]), float(g[f"{name}_scale"]))
        want = torch.from_numpy(g[f"{name}_out"])
        pinned = g[f"{name}_ref_pinned"]
        for b, is_pinned in enumerate(pinned):
            n_q, n_k = cu_q[b + 1] - cu_q[b], cu_k[b + 1] - cu_k[b]
            assert bool(is_pinned) == bool(n_q == n_k or n_q == 1), (name, b)
            rows = slice(cu_q[b], cu_q[b + 1])
            assert (o[rows] - want[rows]).abs().max().item() <= 1e-5, (name, b, "pinned" if is_pinned else "unpinned")
            pinned_seqs += int(is_pinned)
            unpinned_seqs += int(not is_pinned)
    assert pinned_seqs >= 30 and unpinned_seqs >= 3
    names = set(g["names"].tolist())
    assert {"bs256_full_prefill_g2", "bs256_full_prefill_g4", "bs256_full_prefill_g8", "bs256_decode_g2",
            "bs256_decode_g4", "bs256_decode_g8"} <= names


FULL_PREFILL_CASES = ["full_prefill_g2", "bs256_full_prefill_g2", "bs256_full_prefill_g4", "bs256_full_prefill_g8"]


def chunk_problems(g, name):
    """Every full-prefill sequence of fixture case `name` cut into chunks (uneven, one of a single row):
    yields (chunk problem as tensors, rows of the case's stored output the chunk must reproduce)."""
    cu_q = g[f"{name}_cu_q"]
    q = bf16(g[f"{name}_q"])
    bt = torch.from_numpy(g[f"{name}_block_tables"])
    for b in range(len(cu_q) - 1):
        n = int(cu_q[b + 1] - cu_q[b])
        if n < 4:
            continue
        cuts = sorted({0, n // 3, n // 3 + 1, (2 * n) // 3, n})
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            rows = slice(int(cu_q[b]) + lo, int(cu_q[b]) + hi)
            yield dict(q=q[rows], cu_q=torch.tensor([0, hi - lo], dtype=torch.int32),
                       cu_k=torch.tensor([0, hi], dtype=torch.int32), block_tables=bt[b:b + 1].contiguous(),
                       max_q=hi - lo, max_k=hi), rows


def test_chunked_prefill_reproduces_reference_pinned_full_prefill_rows(golden):
    """Anchor for the bottom-right causal alignment that needs no flash-attn: rows lo..hi-1 of a prompt
    attend to keys 0..row whether the prompt is prefilled at once or the rows arrive as a later chunk
    (n_q = hi - lo queries over n_k = hi cached keys), so a chunk must reproduce those rows of the
    full-prefill output.  The full-prefill outputs stored in g4 are the reference's own dense helper's
    (nanovllm/layers/attention.py:273-335, valid for n_q == n_k); a top-left aligned mask (that helper's
    `tril`, wrong for chunks) would fail here on every chunk with lo > 0."""
    g = golden("g4_attention.npz")
    chunks = 0
    for name in FULL_PREFILL_CASES:
        assert g[f"{name}_ref_pinned"].all()
        want = torch.from_numpy(g[f"{name}_out"])
        kc, vc, scale = bf16(g[f"{name}_k_cache"]), bf16(g[f"{name}_v_cache"]), float(g[f"{name}_scale"])
        for p, rows in chunk_problems(g, name):
            o = ref_attention.paged_attention_ref(p["q"], kc, vc, p["cu_q"], p["cu_k"], p["block_tables"], scale)
            assert (o - want[rows]).abs().max().item() <= 1e-5, (name, rows)
            chunks += 1
    assert chunks >= 20


def test_attention_bottom_right_mask_small_case():
    """n_q=2, n_k=3: query 0 sees keys 0..1, query 1 sees keys 0..2; and decode == no mask."""
    D = 128
    q = torch.zeros(2, 1, D).bfloat16()
    k_cache = torch.zeros(1, 4, 1, D).bfloat16()
    v_cache = torch.zeros(1, 4, 1, D).bfloat16()
    v_cache[0, 0, 0, 0], v_cache[0, 1, 0, 0], v_cache[0, 2, 0, 0] = 1.0, 2.0, 6.0
    o = ref_attention.paged_attention_ref(q, k_cache, v_cache, torch.tensor([0, 2], dtype=torch.int32),
                                          torch.tensor([0, 3], dtype=torch.int32),
                                          torch.tensor([[0]], dtype=torch.int32), 1.0)
    assert abs(o[0, 0, 0].item() - 1.5) < 1e-6       # mean of keys 0,1
    assert abs(o[1, 0, 0].item() - 3.0) < 1e-6       # mean of keys 0,1,2


def test_attention_rows_without_keys_are_zero():
    D = 128
    q = torch.ones(3, 1, D).bfloat16()
    k_cache = torch.ones(1, 4, 1, D).bfloat16()
    v_cache = torch.ones(1, 4, 1, D).bfloat16()
    # n_q=3 > n_k=1: queries 0 and 1 see nothing (bottom-right alignment), query 2 sees key 0
    o = ref_attention.paged_attention_ref(q, k_cache, v_cache, torch.tensor([0, 3], dtype=torch.int32),
                                          torch.tensor([0, 1], dtype=torch.int32),
                                          torch.tensor([[0]], dtype=torch.int32), 1.0)
    assert o[0].abs().max() == 0 and o[1].abs().max() == 0 and abs(o[2, 0, 0].item() - 1.0) < 1e-6


def test_block_hash_known_answers():
    import xxhash
    with open(os.path.join(GOLDEN, "g5_hash.json")) as f:
        vectors = json.load(f)
    assert vectors[0]["hash"] == 2877822695146591398 and vectors[1]["hash"] == 17010504966165004578
    for v in vectors:
        h = xxhash.xxh64()
        if v["prev"] != -1:
            h.update(int(v["prev"]).to_bytes(8, "little"))
        h.update(np.array(v["tokens"], dtype=np.int32).tobytes())
        assert h.intdigest() == v["hash"]


class _Seq:
    def __init__(self, prompt, outputs, computed, scheduled, table):
        self.prompt_tokens, self.output_tokens = prompt, outputs
        self.num_computed_tokens, self.num_scheduled_tokens, self.block_table = computed, scheduled, table

    @property
    def is_prefill(self):
        return len(self.output_tokens) == 0

    @property
    def num_tokens(self):
        return len(self.prompt_tokens) + len(self.output_tokens)


def test_metadata_builder_integers():
    """Hand-computed expectations for nanovllm/engine/model_runner.py:56-119 (block_size 4)."""
    a = _Seq(list(range(100, 110)), [], 4, 6, [7, 2, 5])          # chunked prefill, positions 4..9
    b = _Seq([1, 2, 3], [50, 51], 4, 1, [9, 0])                    # decode: last output at position 4
    m = ref_metadata.prepare_inputs_ref([a, b], 4)
    assert m["input_ids"] == [104, 105, 106, 107, 108, 109, 51]
    assert m["positions"] == [4, 5, 6, 7, 8, 9, 4]
    assert m["slot_mapping"] == [2 * 4 + 0, 2 * 4 + 1, 2 * 4 + 2, 2 * 4 + 3, 5 * 4 + 0, 5 * 4 + 1, 0 * 4 + 0]
    assert m["cu_seqlens_q"] == [0, 6, 7] and m["cu_seqlens_k"] == [0, 10, 15]
    assert m["max_seqlen_q"] == 6 and m["max_seqlen_k"] == 10
    assert m["block_tables"] == [[7, 2, 5], [9, 0, -1]]


LLAMA3 = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
              original_max_position_embeddings=8192)


def test_llama3_rope_rule_known_bands():
    """The reference has no rope scaling (nanovllm/layers/rope.py:67), so the oracle's llama3 rule is
    pinned by its published definition: short waves untouched, long waves divided by `factor`, the
    band in between monotone and inside the two; and, where the installed transformers package
    offers its llama3 initialiser, by agreement with that independent implementation."""
    base = 1.0 / (5e5 ** (torch.arange(0, 128, 2, dtype=torch.int64).float() / 128))
    got = ref_layers.llama3_inv_freq_ref(base, LLAMA3)
    wavelen = 2 * np.pi / base.numpy()
    short, long_ = wavelen < 8192 / 4.0, wavelen > 8192 / 1.0
    assert short.any() and long_.any() and (~short & ~long_).any()
    assert np.array_equal(got.numpy()[short], base.numpy()[short])
    assert np.array_equal(got.numpy()[long_], (base / 8.0).numpy()[long_])
    mid = ~short & ~long_
    assert ((got.numpy()[mid] <= base.numpy()[mid]) & (got.numpy()[mid] >= (base / 8.0).numpy()[mid])).all()
    assert (np.diff(got.numpy()) < 0).all()                    # still strictly decreasing in frequency index
    try:
        from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS

        class Cfg:
            rope_theta = 5e5
            hidden_size = 4096
            num_attention_heads = 32
            head_dim = 128
            partial_rotary_factor = 1.0
            max_position_embeddings = 131072
            rope_scaling = LLAMA3
            rope_parameters = dict(LLAMA3, rope_theta=5e5)

            def standardize_rope_params(self):
                pass
        want, _ = ROPE_INIT_FUNCTIONS["llama3"](Cfg(), "cpu")
    except Exception:
        return                                                   # helper absent / other signature: bands above stand
    assert torch.allclose(got, want, rtol=1e-6, atol=0)


def test_llama3_table_of_the_product_equals_the_oracle_table():
    """Host-side table builder of the product (layers/rope.py) against the oracle's independent
    scalar restatement: identical fp32 bits, including rows far beyond the original 8192 positions."""
    from nanovllm_amd.layers.rope import build_cos_sin_cache
    mine = build_cos_sin_cache(128, 131072, 5e5, rope_scaling=LLAMA3)
    ref = ref_layers.build_cos_sin_cache(128, 131072, 5e5, LLAMA3)
    assert mine.shape == ref.shape == (131072, 1, 128)
    assert torch.equal(mine.view(torch.int32), ref.view(torch.int32))
    plain = ref_layers.build_cos_sin_cache(128, 64, 5e5)
    assert not torch.equal(plain[63], ref[63])                   # the scaling does change the table
