"""Oracle (test infrastructure): torch modules with the layer contract, computing with the CPU
restatements of oracle/ref_layers.py and oracle/ref_attention.py.  Device-agnostic torch code, so
the same modules can also run on GPU tensors to give an eager reference that shares the GEMMs with
the HIP path.  Used only by tests / smoke to exercise host plumbing and token parity; the product
model never imports this module."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ref_attention, ref_layers


def _get_context():
    import nanovllm_amd
    return nanovllm_amd.get_context()


class Attention(nn.Module):
    def __init__(self, num_heads, head_dim, num_kv_heads, attn_scale, max_seq_len):
        super().__init__()
        assert num_heads % num_kv_heads == 0
        self.num_heads, self.head_dim, self.num_kv_heads = num_heads, head_dim, num_kv_heads
        self.attn_scale, self.max_seq_len = attn_scale, max_seq_len
        self.k_cache = None
        self.v_cache = None

    def attend(self, q):
        ctx = _get_context()
        o = ref_attention.paged_attention_ref(q, self.k_cache, self.v_cache, ctx.cu_seqlens_q, ctx.cu_seqlens_k,
                                              ctx.block_tables, self.attn_scale)
        return o.to(q.dtype)

    def forward(self, q, k, v):
        ctx = _get_context()
        ref_attention.store_kvcache_ref(k, v, self.k_cache, self.v_cache, ctx.slot_mapping, self.k_cache.shape[1])
        return self.attend(q)


class RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6, dtype=None, device=None):
        super().__init__()
        self.hidden_size, self.eps = hidden_size, eps
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=dtype, device=device))

    def forward(self, x, residual=None):
        return ref_layers.rmsnorm_ref(x, self.weight, self.eps, residual)


class Rope(nn.Module):
    def __init__(self, head_size, max_position, base, rope_scaling=None):
        super().__init__()
        # rope_scaling: None as in the reference (rope.py:67), or a llama3 dict (oracle's own restatement)
        self.register_buffer("cos_sin_cache",
                             ref_layers.build_cos_sin_cache(head_size, max_position, base, rope_scaling),
                             persistent=False)

    def forward(self, positions, query, key):
        return ref_layers.rope_ref(positions, query, key, self.cos_sin_cache)


class SiluAndMul(nn.Module):
    def forward(self, x):
        return ref_layers.silu_and_mul_ref(x)


_ropes: dict = {}


def make_rope(head_size, max_position, base, rope_scaling=None):
    key = (head_size, max_position, base, tuple(sorted(rope_scaling.items())) if rope_scaling else None)
    if key not in _ropes:
        _ropes[key] = Rope(head_size, max_position, base, rope_scaling)
    return _ropes[key]
