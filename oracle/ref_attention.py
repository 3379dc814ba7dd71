"""Oracle (test infrastructure): KV-cache scatter and varlen paged attention on the CPU.

store_kvcache_ref follows Attention.store_kvcache, nanovllm/layers/attention.py:210-219.
paged_attention_ref restates the call at nanovllm/layers/attention.py:241-252,
``flash_attn_varlen_func(q, k_cache, v_cache, cu_seqlens_q, cu_seqlens_k, max_seqlen_q,
max_seqlen_k, softmax_scale, causal=True, block_table=...)``.  flash-attn is a third-party
dependency that is not vendored, not version-pinned and not installed (parity unpinned); its
published contract is restated here: keys of sequence b are gathered through block_table[b],
query head h uses kv head h // (H_q / H_kv), the causal mask is aligned to the bottom-right
corner (query i of n_q sees keys j <= n_k - n_q + i), rows with no visible key give 0,
arithmetic in fp32 on the bf16 inputs.
"""
from __future__ import annotations

import math

import torch


def store_kvcache_ref(k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                      slot_mapping: torch.Tensor, block_size: int) -> None:
    """attention.py:214-219: cache[slot // bs, slot % bs] = row (in place)."""
    sm = slot_mapping.long()
    bi = sm // block_size
    bo = sm % block_size
    k_cache[bi, bo] = k
    v_cache[bi, bo] = v


def gather_kv(cache: torch.Tensor, block_table_row: torch.Tensor, n_k: int) -> torch.Tensor:
    """Keys 0..n_k-1 of one sequence: key j lives at cache[block_table[j // bs], j % bs]."""
    bs = cache.shape[1]
    j = torch.arange(n_k, device=cache.device)
    blocks = block_table_row.long()[j // bs]
    return cache[blocks, j % bs]                        # [n_k, H_kv, D]


def paged_attention_ref(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                        cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor,
                        block_tables: torch.Tensor, scale: float, causal: bool = True) -> torch.Tensor:
    """Returns fp32 [T, H_q, D]."""
    T, H_q, D = q.shape
    H_kv = k_cache.shape[2]
    G = H_q // H_kv
    dev = q.device
    out = torch.zeros(T, H_q, D, dtype=torch.float32, device=dev)
    cq = cu_seqlens_q.tolist()
    ck = cu_seqlens_k.tolist()
    for b in range(len(cq) - 1):
        n_q = cq[b + 1] - cq[b]
        n_k = ck[b + 1] - ck[b]
        if n_q == 0 or n_k == 0:
            continue
        qb = q[cq[b]:cq[b + 1]].float().transpose(0, 1)                       # [H_q, n_q, D]
        kb = gather_kv(k_cache, block_tables[b], n_k).float()                 # [n_k, H_kv, D]
        vb = gather_kv(v_cache, block_tables[b], n_k).float()
        kb = kb.repeat_interleave(G, dim=1).transpose(0, 1)                   # [H_q, n_k, D]
        vb = vb.repeat_interleave(G, dim=1).transpose(0, 1)
        s = torch.matmul(qb, kb.transpose(1, 2)) * scale                      # [H_q, n_q, n_k]
        if causal:
            i = torch.arange(n_q, device=dev).unsqueeze(1)
            j = torch.arange(n_k, device=dev).unsqueeze(0)
            visible = j <= i + (n_k - n_q)
            s = s.masked_fill(~visible, float("-inf"))
            dead = ~visible.any(dim=1)                                        # rows seeing nothing
        else:
            dead = torch.zeros(n_q, dtype=torch.bool, device=dev)
        s[:, dead, :] = 0.0
        p = torch.softmax(s, dim=-1)
        o = torch.matmul(p, vb)                                               # [H_q, n_q, D]
        o[:, dead, :] = 0.0
        out[cq[b]:cq[b + 1]] = o.transpose(0, 1)
    return out


def attention_forward_ref(q, k, v, k_cache, v_cache, slot_mapping, cu_seqlens_q, cu_seqlens_k,
                          block_tables, scale, block_size):
    """Attention.forward, attention.py:221-254: store first (:230-231), then attend (:241-252)."""
    store_kvcache_ref(k, v, k_cache, v_cache, slot_mapping, block_size)
    return paged_attention_ref(q, k_cache, v_cache, cu_seqlens_q, cu_seqlens_k, block_tables, scale)


def decode_attention_bytes(ctx_lens, H_q: int, H_kv: int, D: int, block_size: int) -> int:
    """Algorithmic HBM bytes of one decode-attention layer pass: K and V read once, Q read and
    O written once, block-table entries and lengths."""
    total = 0
    for c in ctx_lens:
        total += 2 * c * H_kv * D * 2 + 2 * H_q * D * 2 + 4 * math.ceil(c / block_size) + 8
    return total


def prefill_attention_flops(n_q: int, n_k: int, H_q: int, D: int) -> int:
    """Algorithmic flops (QK^T and PV, 2 per MAC) of the visible (query, key) pairs."""
    pairs = n_q * (n_k - n_q) + n_q * (n_q + 1) // 2
    return 4 * D * H_q * pairs
