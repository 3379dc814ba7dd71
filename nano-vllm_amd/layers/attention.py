"""Attention layer over the paged KV cache, backed by gfx950 HIP kernels.

Drop-in for the reference's ``Attention`` (nanovllm/layers/attention.py:7-37, 210-254):
same constructor arguments, the same ``k_cache`` / ``v_cache`` attributes that the model
runner fills with per-layer views ``[num_blocks, block_size, H_kv, D]``
(nanovllm/engine/model_runner.py:186-192), and ``forward(q, k, v) -> o`` reading the step's
``CacheContext``.  Where the reference scatters with ``index_put_`` and then calls
``flash_attn_varlen_func(causal=True, block_table=...)``, this layer runs

  * ``nv_store_kvcache``        -- the slot_mapping scatter (bit-exact),
  * ``nv_attn_decode_paged``    -- sequences with one query token (bandwidth-bound kernel),
  * ``nv_attn_prefill_varlen``  -- sequences with several query tokens (bf16 MFMA kernel),

and a mixed step launches both attention kernels, each skipping the other's sequences.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..utils.cache_context import get_context


def _derive_hints(ctx, num_heads: int = 0, num_kv_heads: int = 0) -> None:
    """A caller that gave no hints (the reference's runner builds its context from host lists and passes
    only the tensors): read the two cumulative-length tensors back ONCE per step -- one small device-to-host
    copy in the first layer of a step that mixes decode rows and prompt chunks; pure decode steps never get
    here -- and keep the result on the context object, which every layer of the step shares.  Without it
    each of the step's layers would launch both attention kernels over all sequences."""
    cu = torch.stack([ctx.cu_seqlens_q, ctx.cu_seqlens_k]).cpu()
    n_q = (cu[0, 1:] - cu[0, :-1]).tolist()
    n_k = (cu[1, 1:] - cu[1, :-1]).tolist()
    nd = 0
    while nd < len(n_q) and n_q[nd] == 1:
        nd += 1
    ctx.num_leading_decode = nd
    ctx.max_seqlen_k_decode = max(n_k[:nd], default=0)
    ctx.max_seqlen_q_rest = max(n_q[nd:], default=0)
    ctx.max_seqlen_k_rest = max(n_k[nd:], default=0)
    ctx.num_decode_seqs = sum(1 for n in n_q if n == 1)
    # the lengths are on the host now: the decode rows' launch can be planned with them (see ops.decode_suggest_chunk)
    if nd and num_heads and num_kv_heads and not getattr(ctx, "decode_chunk_keys", 0):
        ctx.decode_chunk_keys = ops.decode_suggest_chunk(nd, num_heads, num_kv_heads, ctx.max_seqlen_k_decode,
                                                         sum(n_k[:nd]))


class Attention(nn.Module):
    """q: [T, H_q, D], k/v: [T, H_kv, D] (flattened "super sequence"), returns [T, H_q, D]."""

    def __init__(self, num_heads: int, head_dim: int, num_kv_heads: int, attn_scale: float,
                 max_seq_len: int):
        super().__init__()
        assert num_heads % num_kv_heads == 0
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.num_kv_heads = num_kv_heads
        self.attn_scale = attn_scale
        self.max_seq_len = max_seq_len
        self.num_queries_per_kv = num_heads // num_kv_heads
        # bound by the runner after the cache is allocated
        self.k_cache: torch.Tensor | None = None
        self.v_cache: torch.Tensor | None = None

    def store_kvcache(self, k: torch.Tensor, v: torch.Tensor, slot_mapping: torch.Tensor,
                      block_size: int | None = None) -> None:
        """Scatter the step's K/V rows into the cache.  ``block_size`` is accepted for signature
        compatibility; the block size actually used is the cache's own second dimension."""
        if block_size is not None and block_size != self.k_cache.shape[1]:
            raise ValueError(
                f"context block_size {block_size} != cache block size {self.k_cache.shape[1]}")
        ops.store_kvcache(k, v, self.k_cache, self.v_cache, slot_mapping)

    def attend(self, q: torch.Tensor) -> torch.Tensor:
        """Attention of q against the cache as described by the current context (no store)."""
        ctx = get_context()
        B = ctx.cu_seqlens_q.numel() - 1
        chunk = getattr(ctx, "decode_chunk_keys", 0) or 0      # (a context of the reference's class has no such field)
        if ctx.max_seqlen_q <= 1:
            return ops.attn_decode_paged(q, self.k_cache, self.v_cache, ctx.cu_seqlens_k,
                                         ctx.block_tables, ctx.max_seqlen_k, self.attn_scale,
                                         cu_seqlens_q=ctx.cu_seqlens_q, chunk_keys=chunk)
        if ctx.num_leading_decode is None and ctx.num_decode_seqs is None:
            _derive_hints(ctx, self.num_heads, self.num_kv_heads)
            chunk = getattr(ctx, "decode_chunk_keys", 0) or 0
        nd = ctx.num_leading_decode
        if nd is not None:
            # decode rows lead the batch: each kernel gets exactly its own sequences (no idle
            # workgroups, and a launch plan sized from its own longest context)
            o = torch.empty_like(q) if q.is_contiguous() else torch.empty(q.shape, dtype=q.dtype, device=q.device)
            if nd > 0:
                ops.attn_decode_paged(q, self.k_cache, self.v_cache, ctx.cu_seqlens_k[:nd + 1],
                                      ctx.block_tables[:nd], ctx.max_seqlen_k_decode, self.attn_scale,
                                      cu_seqlens_q=ctx.cu_seqlens_q[:nd + 1], out=o, chunk_keys=chunk)
            if nd < B:
                ops.attn_prefill_varlen(q, self.k_cache, self.v_cache, ctx.cu_seqlens_q[nd:], ctx.cu_seqlens_k[nd:],
                                        ctx.block_tables[nd:], ctx.max_seqlen_q_rest, ctx.max_seqlen_k_rest,
                                        self.attn_scale, skip_single_query=False, out=o)
            return o
        n_dec = ctx.num_decode_seqs
        mixed = n_dec is None or n_dec > 0
        if n_dec is not None and n_dec == B:  # max_seqlen_q was an over-estimate
            return ops.attn_decode_paged(q, self.k_cache, self.v_cache, ctx.cu_seqlens_k,
                                         ctx.block_tables, ctx.max_seqlen_k, self.attn_scale,
                                         cu_seqlens_q=ctx.cu_seqlens_q)
        o = ops.attn_prefill_varlen(q, self.k_cache, self.v_cache, ctx.cu_seqlens_q, ctx.cu_seqlens_k,
                                    ctx.block_tables, ctx.max_seqlen_q, ctx.max_seqlen_k,
                                    self.attn_scale, skip_single_query=mixed)
        if mixed:
            ops.attn_decode_paged(q, self.k_cache, self.v_cache, ctx.cu_seqlens_k, ctx.block_tables,
                                  ctx.max_seqlen_k, self.attn_scale, cu_seqlens_q=ctx.cu_seqlens_q, out=o)
        return o

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        ctx = get_context()
        if self.k_cache is None or self.v_cache is None or self.k_cache.numel() == 0:
            raise RuntimeError(
                "Attention.k_cache / v_cache are not bound: this implementation always attends "
                "through the paged cache (bind per-layer views before the first forward)")
        self.store_kvcache(k, v, ctx.slot_mapping, ctx.block_size)
        return self.attend(q)
