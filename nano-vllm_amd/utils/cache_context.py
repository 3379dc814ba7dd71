"""Per-step attention metadata carrier.

Same contract as the reference's ``nanovllm/utils/cache_context.py:6-39``: a dataclass held in
a ``ContextVar``; ``Attention.forward`` fetches it with ``get_context()``, which raises
``RuntimeError`` when no context was set.  Field names and defaults match the reference; two
optional hints are added for this implementation (``num_decode_seqs``, ``num_leading_decode`` and
its bounds) and may be left unset by callers that do not know them.
"""
from __future__ import annotations

from contextvars import ContextVar
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class CacheContext:
    layer_idx: int = -1
    is_prefill: bool = False
    # [T] int32: flat slot (block * block_size + offset) where each new token's K/V is stored
    slot_mapping: Optional[torch.Tensor] = None
    # [B+1] int32 cumulative query / key lengths of the flattened batch
    cu_seqlens_q: Optional[torch.Tensor] = None
    cu_seqlens_k: Optional[torch.Tensor] = None
    max_seqlen_q: int = 0
    max_seqlen_k: int = 0
    # [B, max_blocks] int32 physical block ids, right-padded with -1
    block_tables: Optional[torch.Tensor] = None
    context_lens: Optional[torch.Tensor] = None

    attention_mask: Optional[torch.Tensor] = None
    block_size: int = 256

    # Extension (host-side hint, optional): how many sequences of the step have exactly one
    # query token.  None = unknown; the attention layer then launches both of its kernels
    # whenever max_seqlen_q > 1 and lets each skip the rows that are not its regime.
    num_decode_seqs: Optional[int] = None
    # Extension (optional): the first `num_leading_decode` sequences of the batch have exactly one
    # query token each (schedulers put running sequences first), with `max_seqlen_k_decode` the
    # longest of their contexts; `max_seqlen_q_rest` / `max_seqlen_k_rest` bound the sequences after
    # them.  When given, a mixed step launches the decode kernel over the leading rows only and the
    # prefill kernel over the rest, each planned from its own rows.
    num_leading_decode: Optional[int] = None
    max_seqlen_k_decode: int = 0
    max_seqlen_q_rest: int = 0
    max_seqlen_k_rest: int = 0
    # Extension (optional): keys per decode work item chosen by a host that knows the batch's lengths
    # (`ops.decode_suggest_chunk`: a few long sequences among many short ones); 0 = the launch plan decides
    # from `max_seqlen_k` alone.
    decode_chunk_keys: int = 0


_cache_context: ContextVar[Optional[CacheContext]] = ContextVar("CacheContext", default=None)


def set_context(ctx: Optional[CacheContext]) -> None:
    """Install the context for the coming forward pass."""
    _cache_context.set(ctx)


def get_context() -> CacheContext:
    """Return the current context; RuntimeError if none was set."""
    ctx = _cache_context.get()
    if ctx is None:
        raise RuntimeError("CacheContext not set. Call set_context() before forward pass.")
    return ctx
