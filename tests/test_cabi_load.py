"""CPU: the C-ABI library loads and exports every symbol the header declares; host-side module
contract (constructors, context errors, refusal of CPU tensors).  No compute calls."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT

import nanovllm_amd
from nanovllm_amd import _lib, ops


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nanovllm_hip.h")).read()
    return re.findall(r"NV_API\s+[\w\s\*]+?\b(nv_\w+)\s*\(", text)


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    assert len(syms) == len(set(syms)) >= 11
    assert set(syms) == set(_lib.PROTOTYPES), "ctypes prototypes and header disagree"


def test_library_loads_and_exports_every_declared_symbol():
    dll = _lib.load()
    for name in declared_symbols():
        assert hasattr(dll, name), f"{name} not exported"
    assert dll.nv_abi_version() == _lib.ABI_VERSION
    assert dll.nv_strerror(0) == b"ok" and dll.nv_strerror(-3) == b"workspace too small"


def test_argument_errors_are_reported_without_a_gpu():
    dll = _lib.lib()
    # head_dim 64 is unsupported; validation happens before any device access
    assert dll.nv_attn_decode_workspace_bytes(4, 16, 8, 64, 100) == 0
    rc = dll.nv_attn_decode_paged(None, None, None, None, None, None, None, 4, 1, 16, 16, 8, 64, 1, 16, 1.0,
                                  2048, 2048, None, 0, None)
    assert rc == -2 and b"head_dim" in dll.nv_last_error()
    rc = dll.nv_store_kvcache(None, None, None, None, None, 3, 8, 128, 4, 256, 1024, 1024, None)
    assert rc == -1 and b"null" in dll.nv_last_error()
    rc = dll.nv_rmsnorm(None, None, None, 0, 1024, 1024, 1e-6, None)      # zero rows: a no-op
    assert rc == 0
    with pytest.raises(_lib.NanoVllmHipError):
        _lib.check(-1, "x")


def test_header_and_library_from_plain_c(tmp_path):
    """The header is C (not only C++) and the library binds from a plain C program via dlopen."""
    import subprocess
    exe = str(tmp_path / "cabi_from_c")
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cabi_from_c.c"), "-o", exe, "-ldl"], check=True)
    out = subprocess.run([exe, _lib.LIB_PATH], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.returncode, out.stdout, out.stderr)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load(str(tmp_path / "nope.so"))


def test_context_contract():
    nanovllm_amd.set_context(None)
    with pytest.raises(RuntimeError, match="CacheContext not set"):
        nanovllm_amd.get_context()
    ctx = nanovllm_amd.CacheContext()
    assert ctx.block_size == 256 and ctx.layer_idx == -1 and ctx.is_prefill is False
    assert ctx.slot_mapping is None and ctx.max_seqlen_q == 0 and ctx.block_tables is None
    nanovllm_amd.set_context(ctx)
    assert nanovllm_amd.get_context() is ctx
    nanovllm_amd.set_context(None)


def test_hints_are_derived_once_from_the_reference_context_fields():
    """A context as the reference's runner builds it (nanovllm/engine/model_runner.py:102-111: tensors and
    the two maxima, no hints): the attention layer derives which leading sequences are decode rows and the
    per-kernel bounds from cu_seqlens_q / cu_seqlens_k, and keeps them on the context for the step's other layers."""
    from nanovllm_amd.layers.attention import _derive_hints
    i32 = lambda xs: torch.tensor(xs, dtype=torch.int32)
    ctx = nanovllm_amd.CacheContext(cu_seqlens_q=i32([0, 1, 2, 132, 133, 645]), cu_seqlens_k=i32([0, 300, 317, 447, 1447, 2471]),
                                    max_seqlen_q=512, max_seqlen_k=1024)
    _derive_hints(ctx)
    assert ctx.num_leading_decode == 2 and ctx.num_decode_seqs == 3
    assert ctx.max_seqlen_k_decode == 300 and ctx.max_seqlen_q_rest == 512 and ctx.max_seqlen_k_rest == 1024
    ctx = nanovllm_amd.CacheContext(cu_seqlens_q=i32([0, 7, 8]), cu_seqlens_k=i32([0, 7, 40]), max_seqlen_q=7, max_seqlen_k=33)
    _derive_hints(ctx)
    assert ctx.num_leading_decode == 0 and ctx.max_seqlen_k_decode == 0 and ctx.max_seqlen_q_rest == 7


def test_module_signatures_and_parameters():
    attn = nanovllm_amd.Attention(16, 128, 8, 128 ** -0.5, 4096)
    assert attn.num_queries_per_kv == 2 and attn.k_cache is None and attn.v_cache is None
    with pytest.raises(AssertionError):
        nanovllm_amd.Attention(10, 128, 4, 1.0, 16)
    norm = nanovllm_amd.RMSNorm(1024, eps=1e-6, dtype=torch.bfloat16)
    assert norm.weight.shape == (1024,) and norm.weight.dtype == torch.bfloat16
    assert [n for n, _ in norm.named_parameters()] == ["weight"]
    rope = nanovllm_amd.RotaryPositionEmbedding(128, 128, 64, 1e6)
    assert rope.cos_sin_cache.shape == (64, 1, 128) and rope.cos_sin_cache.dtype == torch.float32
    assert "cos_sin_cache" not in rope.state_dict()          # non-persistent buffer
    with pytest.raises(AssertionError):
        nanovllm_amd.RotaryPositionEmbedding(128, 64, 64, 1e6)
    nanovllm_amd.get_rope.cache_clear()
    assert nanovllm_amd.get_rope(128, 128, 32, 1e6) is nanovllm_amd.get_rope(128, 128, 32, 1e6)
    with pytest.raises(AssertionError):
        nanovllm_amd.get_rope(128, 128, 32, 1e6, rope_scaling=("llama3",))


def test_rope_table_matches_oracle():
    from oracle.ref_layers import build_cos_sin_cache
    rope = nanovllm_amd.RotaryPositionEmbedding(128, 128, 512, 1e6)
    assert torch.equal(rope.cos_sin_cache, build_cos_sin_cache(128, 512, 1e6))


def test_ops_refuse_cpu_tensors():
    x = torch.zeros(4, 1024, dtype=torch.bfloat16)
    w = torch.ones(1024, dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="GPU tensor"):
        ops.rmsnorm(x, w, 1e-6)
    attn = nanovllm_amd.Attention(4, 128, 2, 1.0, 64)
    nanovllm_amd.set_context(nanovllm_amd.CacheContext())
    with pytest.raises(RuntimeError, match="not bound"):
        attn(torch.zeros(1, 4, 128), torch.zeros(1, 2, 128), torch.zeros(1, 2, 128))
    nanovllm_amd.set_context(None)


def test_fast_division_by_launch_constants_is_exact():
    """The kernels take a block index apart with host-prepared divisors (csrc/nv_common.h: make_fastdiv /
    fast_div): shr = 31 + ceil(log2 d), mul = ceil(2^shr / d), n / d = (n * mul) >> shr.  The same arithmetic
    restated here must be exact for every 0 <= n < 2^31 that matters: all small divisors exhaustively over small
    n, and edge values (multiples of d and their neighbours, 2^31 - 1) for small, power-of-two, odd and large
    divisors; mul must fit 32 bits (the device multiplies 32 x 32 -> 64)."""
    import random

    def make(d):
        l = 0
        while (1 << l) < d:
            l += 1
        shr = 31 + l
        return ((1 << shr) + d - 1) // d, shr

    rnd = random.Random(0)
    divisors = list(range(1, 300)) + [2 ** k + e for k in range(8, 31) for e in (-1, 0, 1)] + \
        [rnd.randrange(1, 2 ** 31) for _ in range(200)]
    for d in divisors:
        if not 1 <= d < 2 ** 31:
            continue
        mul, shr = make(d)
        assert mul < 2 ** 32 and shr < 64, (d, mul, shr)
        ns = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 2 ** 31 - 1, 2 ** 31 - d, (2 ** 31 - 1) // d * d,
              (2 ** 31 - 1) // d * d - 1] + [rnd.randrange(0, 2 ** 31) for _ in range(50)]
        if d < 300:
            ns += list(range(0, 4 * d + 2))
        for n in ns:
            if 0 <= n < 2 ** 31:
                assert (n * mul) >> shr == n // d, (n, d)


def test_decode_chunk_suggestion_known_answers():
    """nv_attn_decode_suggest_chunk is host arithmetic (no GPU): balanced batches and launches with few
    (sequence, kv head) pairs keep the plan's choice (0); a few long sequences among many short ones get pieces of
    about three quarters of a mean sequence, at most 32 ranges per sequence."""
    dll = _lib.lib()
    f = dll.nv_attn_decode_suggest_chunk
    assert f(256, 16, 8, 767, 256 * 640) == 0                      # the benchmark batch: longest = 1.2 x mean
    assert f(256, 16, 8, 2048, 256 * 2048) == 0                    # uniform long sequences: no cut wanted
    assert f(256, 16, 8, 1023, 256 * 640) == 0                     # 1.6 x the mean but only 383 keys beyond it
    # twice the resident work items, no sequence outlasts the rest: nothing to cut; lengths that spread out get the
    # plan's own cap back, which cuts nothing and selects the prefetching kernel
    assert f(512, 16, 8, 2000, 512 * 1050) == 2048
    assert f(512, 16, 8, 767, 512 * 640) == 0 and f(512, 16, 8, 2000, 512 * 1900) == 0
    assert f(512, 32, 8, 2000, 512 * 1050) == 0                    # the MFMA-score kernels always prefetch
    assert f(512, 16, 8, 6000, 511 * 1050 + 6000) == 0             # ... nor does one of 6000 keys among them
    assert f(64, 16, 8, 2000, 64 * 1050) == 0                      # few pairs, 1.9 x the mean
    assert f(64, 16, 8, 1500, 63 * 640 + 1500) == 0                # few pairs, 2.3 x the mean
    assert f(64, 16, 8, 3000, 63 * 640 + 3000) > 0                 # few pairs, 4.4 x the mean
    assert f(256, 16, 8, 1500, 255 * 640 + 1500) == 512            # resident at once: ends with its longest sequence
    assert f(1024, 16, 8, 4096, 1023 * 640 + 4096) == 0            # 6 x the mean, but 1/5 of a compute unit's share
    assert f(512, 16, 8, 8192, 511 * 640 + 8192) == 512            # 12 x the mean and most of a compute unit's share
    # few (sequence, kv head) pairs: the plan cuts every sequence max / chunk ways to fill the launch, which only
    # equal lengths do; with the sum known the chunk is what fills it
    assert f(8, 32, 8, 32768, 8 * 32768) == 0 and f(1, 32, 8, 32768, 32768) == 0
    assert f(8, 32, 8, 32768, 7 * 1024 + 32768) == 320
    assert f(32, 16, 8, 16384, 31 * 500 + 16384) == 256
    assert f(257, 16, 8, 8192, 256 * 640 + 8192) == 512            # mean 670 -> 0.75 x, rounded up to 64
    assert f(257, 16, 8, 32768, 256 * 640 + 32768) == 1024         # 576 would need 57 ranges: 32 at most
    assert f(264, 16, 8, 2048, 256 * 256 + 8 * 2048) == 256        # never below 256 keys
    assert f(257, 64, 8, 8192, 256 * 640 + 8192) == 1024           # the MFMA-score kernels: pieces twice as long
    assert f(0, 16, 8, 100, 100) == 0 and f(4, 16, 8, 0, 0) == 0 and f(4, 16, 5, 100, 100) == 0
    # the workspace of a hinted launch covers its ranges; an unsplit launch needs the 16-byte minimum
    w = dll.nv_attn_decode_workspace_bytes_chunked
    assert w(257, 16, 8, 128, 8192, 0) == dll.nv_attn_decode_workspace_bytes(257, 16, 8, 128, 8192)
    assert w(257, 16, 8, 128, 8192, 512) > w(257, 16, 8, 128, 8192, 0) > 16
    assert w(256, 16, 8, 128, 767, 0) == 16 and w(256, 16, 8, 128, 500, 512) == 16
    # a chunk is rounded up to 64 keys and never gives more than 256 ranges
    assert w(4, 16, 8, 128, 65536, 64) == w(4, 16, 8, 128, 65536, 256) < w(4, 16, 8, 128, 65536, 0) * 9


def test_derived_hints_include_the_decode_chunk_of_a_skewed_mixed_step():
    """Once the lengths are on the host (a mixed step of a runner that gave no hints), the decode rows' launch is
    planned with them: 256 decode rows of ~640 keys and one of 8192, followed by a prompt chunk."""
    from nanovllm_amd.layers.attention import _derive_hints
    n_k = [640] * 256 + [8192] + [512]
    n_q = [1] * 257 + [512]
    cu = lambda xs: torch.tensor([0] + torch.tensor(xs).cumsum(0).tolist(), dtype=torch.int32)
    ctx = nanovllm_amd.CacheContext(cu_seqlens_q=cu(n_q), cu_seqlens_k=cu(n_k), max_seqlen_q=512, max_seqlen_k=8192)
    _derive_hints(ctx, 16, 8)
    assert ctx.num_leading_decode == 257 and ctx.max_seqlen_k_decode == 8192 and ctx.decode_chunk_keys == 512
    even = nanovllm_amd.CacheContext(cu_seqlens_q=cu(n_q), cu_seqlens_k=cu([640] * 257 + [512]), max_seqlen_q=512,
                                     max_seqlen_k=640)
    _derive_hints(even, 16, 8)
    assert even.decode_chunk_keys == 0
