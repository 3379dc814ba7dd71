"""Tensor-level wrappers over the C ABI: argument checking, output allocation, stream lookup.

torch is used here only as the owner of device memory and streams; every function hands raw
``data_ptr()`` values to libnanovllm_hip.so on torch's current stream.  Inputs must live on a
GPU; nothing here computes on the host.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

_BF16 = torch.bfloat16
_workspaces: dict[tuple[int, int], torch.Tensor] = {}
_retired: list[torch.Tensor] = []   # outgrown workspaces stay alive: captured graphs may point at them


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need(cond: bool, msg: str) -> None:
    if not cond:
        raise ValueError(msg)


def _gpu_bf16(t: torch.Tensor, name: str) -> None:
    _need(t.is_cuda, f"{name} must be a GPU tensor (this path has no CPU implementation)")
    _need(t.dtype == _BF16, f"{name} must be bfloat16, got {t.dtype}")


def _i32(t: torch.Tensor, name: str) -> None:
    _need(t.is_cuda and t.dtype == torch.int32 and t.is_contiguous(), f"{name} must be a contiguous int32 GPU tensor")


def _rows_ok(t: torch.Tensor, name: str) -> None:
    """[T, H, D] with (H, D) contiguous; the token stride may be larger (slice of a fused tensor)."""
    _need(t.dim() == 3, f"{name} must be [tokens, heads, head_dim]")
    _need(t.stride(2) == 1 and t.stride(1) == t.shape[2], f"{name}: heads and head_dim must be contiguous")


_small_decode_ws: dict[tuple[int, int], torch.Tensor] = {}
_SMALL_DECODE_PAIRS = 8            # (sequence, kv head) pairs up to which a decode launch combines in-launch
_COUNTERS_ZERO = 1                 # NV_DECODE_WS_COUNTERS_ZERO


def _decode_workspace(device: torch.device, nbytes: int, pairs: int) -> tuple[torch.Tensor, int]:
    """Workspace and flags of a decode launch.  Launches with at most 8 (sequence, kv head) pairs get a buffer of
    their own per (device, stream) whose arrival counters start at zero and that nothing else ever writes (the
    shared buffer below also holds the prefill launches' states): the library can then combine the key ranges
    inside the launch instead of in a second kernel, and leaves the counters at zero again."""
    if pairs > _SMALL_DECODE_PAIRS or nbytes <= 16:
        return _workspace(device, nbytes), 0
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(device).cuda_stream)
    ws = _small_decode_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            # zeroing a fresh buffer here would put a 16 MiB memset into the graph: such a capture keeps the merge
            # launch (captures on a stream that has run the step before, as the engine's do, find their buffer)
            return _workspace(device, nbytes), 0
        if ws is not None:
            _retired.append(ws)
        ws = torch.zeros(max(nbytes, 16 << 20), dtype=torch.uint8, device=device)
        _small_decode_ws[key] = ws
    return ws, _COUNTERS_ZERO


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Split-key scratch of the attention launches: one buffer per (device, stream), so launches on different
    streams of one device never share partial states (launches on one stream are ordered)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _retired.append(ws)
        # the first buffer of a device is sized for the usual launches; side streams get what they ask for
        floor = (64 << 20) if not any(k[0] == idx for k in _workspaces) else (1 << 20)
        ws = torch.empty(max(nbytes, floor), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def store_kvcache(k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                  slot_mapping: torch.Tensor) -> None:
    """cache[slot // bs, slot % bs] = row, for K and V (bit-exact)."""
    for t, n in ((k, "k"), (v, "v"), (k_cache, "k_cache"), (v_cache, "v_cache")):
        _gpu_bf16(t, n)
    _rows_ok(k, "k")
    _rows_ok(v, "v")
    _need(k_cache.dim() == 4 and k_cache.is_contiguous() and v_cache.is_contiguous(),
          "caches must be contiguous [num_blocks, block_size, H_kv, D]")
    _need(k_cache.shape == v_cache.shape and k.shape == v.shape, "k/v shape mismatch")
    nb, bs, hkv, d = k_cache.shape
    _need(k.shape[1:] == (hkv, d), "k rows do not match the cache rows")
    _i32(slot_mapping, "slot_mapping")
    _need(slot_mapping.numel() == k.shape[0], "slot_mapping length != number of tokens")
    rc = _lib.lib().nv_store_kvcache(k.data_ptr(), v.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                                     slot_mapping.data_ptr(), k.shape[0], hkv, d, nb, bs,
                                     k.stride(0), v.stride(0), _stream(k))
    _lib.check(rc, "nv_store_kvcache")


def _attn_common(q, k_cache, v_cache, cu_seqlens_k, block_tables):
    _gpu_bf16(q, "q")
    _gpu_bf16(k_cache, "k_cache")
    _gpu_bf16(v_cache, "v_cache")
    _rows_ok(q, "q")
    _need(k_cache.dim() == 4 and k_cache.is_contiguous() and v_cache.is_contiguous()
          and k_cache.shape == v_cache.shape, "caches must be contiguous [num_blocks, block_size, H_kv, D]")
    _i32(cu_seqlens_k, "cu_seqlens_k")
    _need(block_tables.is_cuda and block_tables.dtype == torch.int32 and block_tables.dim() == 2
          and block_tables.is_contiguous(), "block_tables must be a contiguous [B, max_blocks] int32 GPU tensor")
    B = cu_seqlens_k.numel() - 1
    _need(block_tables.shape[0] == B, "block_tables rows != number of sequences")
    return B


def attn_decode_paged(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                      cu_seqlens_k: torch.Tensor, block_tables: torch.Tensor, max_seqlen_k: int,
                      scale: float, cu_seqlens_q: Optional[torch.Tensor] = None,
                      out: Optional[torch.Tensor] = None, chunk_keys: int = 0) -> torch.Tensor:
    """Paged decode attention; sequences with seqlen_q != 1 are left untouched in ``out``.
    `chunk_keys` > 0: the host's choice of keys per work item (`decode_suggest_chunk`), 0 = the plan's."""
    B = _attn_common(q, k_cache, v_cache, cu_seqlens_k, block_tables)
    nb, bs, hkv, d = k_cache.shape
    hq = q.shape[1]
    if cu_seqlens_q is not None:
        _i32(cu_seqlens_q, "cu_seqlens_q")
        _need(cu_seqlens_q.numel() == B + 1, "cu_seqlens_q / cu_seqlens_k length mismatch")
    else:
        _need(q.shape[0] == B, "without cu_seqlens_q there must be one query token per sequence")
    if out is None:
        out = torch.empty((q.shape[0], hq, d), dtype=_BF16, device=q.device)
    else:
        _gpu_bf16(out, "out")
        _rows_ok(out, "out")
        _need(out.shape == q.shape, "out shape != q shape")
    dll = _lib.lib()
    need = dll.nv_attn_decode_workspace_bytes_chunked(B, hq, hkv, d, int(max_seqlen_k), int(chunk_keys))
    ws, flags = _decode_workspace(q.device, need, B * hkv)
    rc = dll.nv_attn_decode_paged_chunked(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(),
                                          cu_seqlens_q.data_ptr() if cu_seqlens_q is not None else None,
                                          cu_seqlens_k.data_ptr(), block_tables.data_ptr(), B,
                                          block_tables.shape[1], int(max_seqlen_k), hq, hkv, d, nb, bs,
                                          float(scale), q.stride(0), out.stride(0), ws.data_ptr(), ws.numel(),
                                          int(chunk_keys), flags, _stream(q))
    _lib.check(rc, "nv_attn_decode_paged_chunked")
    return out


def decode_suggest_chunk(num_seqs: int, hq: int, hkv: int, max_seqlen_k: int, sum_seqlen_k: int) -> int:
    """Keys per decode work item for a batch whose lengths the host knows (0 = the plan's own choice stands):
    a few long sequences among many short ones are cut into pieces of about a mean sequence, so that their
    waves do not end the launch long after the others."""
    return int(_lib.lib().nv_attn_decode_suggest_chunk(int(num_seqs), int(hq), int(hkv), int(max_seqlen_k),
                                                       int(sum_seqlen_k)))


def attn_prefill_varlen(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                        cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor,
                        block_tables: torch.Tensor, max_seqlen_q: int, max_seqlen_k: int, scale: float,
                        skip_single_query: bool = False,
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Varlen (chunked-)prefill attention, bottom-right causal, K/V read through the block table."""
    B = _attn_common(q, k_cache, v_cache, cu_seqlens_k, block_tables)
    nb, bs, hkv, d = k_cache.shape
    hq = q.shape[1]
    _i32(cu_seqlens_q, "cu_seqlens_q")
    _need(cu_seqlens_q.numel() == B + 1, "cu_seqlens_q / cu_seqlens_k length mismatch")
    if out is None:
        out = torch.empty((q.shape[0], hq, d), dtype=_BF16, device=q.device)
    else:
        _gpu_bf16(out, "out")
        _rows_ok(out, "out")
        _need(out.shape == q.shape, "out shape != q shape")
    dll = _lib.lib()
    need = dll.nv_attn_prefill_workspace_bytes(B, q.shape[0], int(max_seqlen_q), int(max_seqlen_k), hq, hkv, d)
    ws = _workspace(q.device, need)
    rc = dll.nv_attn_prefill_varlen(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                                    out.data_ptr(), cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr(),
                                    block_tables.data_ptr(), B, block_tables.shape[1],
                                    int(max_seqlen_q), int(max_seqlen_k), hq, hkv, d, nb, bs,
                                    float(scale), q.stride(0), out.stride(0),
                                    1 if skip_single_query else 0, q.shape[0], ws.data_ptr(), ws.numel(),
                                    _stream(q))
    _lib.check(rc, "nv_attn_prefill_varlen")
    return out


def decode_fused_supported(hq: int, hkv: int, block_size: int) -> bool:
    """Shapes nv_attn_decode_paged_fused takes (others run qknorm_rope_store + attn_decode_paged)."""
    return hq % hkv == 0 and hq // hkv <= 8 and block_size % 16 == 0


def attn_decode_paged_fused(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                            q_weight: Optional[torch.Tensor], k_weight: Optional[torch.Tensor],
                            cos_sin: torch.Tensor, eps: float, slot_mapping: torch.Tensor,
                            k_cache: torch.Tensor, v_cache: torch.Tensor, cu_seqlens_k: torch.Tensor,
                            block_tables: torch.Tensor, max_seqlen_k: int, scale: float,
                            out: Optional[torch.Tensor] = None, chunk_keys: int = 0) -> torch.Tensor:
    """A pure decode step's q/k-norm + rotary + KV scatter + paged attention in one launch (token b is
    sequence b's only new token).  Bit-identical to qknorm_rope_store followed by attn_decode_paged."""
    _rope_common(positions, q, k, cos_sin)
    B = _attn_common(q, k_cache, v_cache, cu_seqlens_k, block_tables)
    nb, bs, hkv, d = k_cache.shape
    hq = q.shape[1]
    _need(q.shape[0] == B, "one query token per sequence")
    _need(k.shape[1:] == (hkv, d), "k rows do not match the cache rows")
    _gpu_bf16(v, "v")
    _rows_ok(v, "v")
    _need(v.shape == k.shape, "k/v shape mismatch")
    _i32(slot_mapping, "slot_mapping")
    _need(slot_mapping.numel() == B, "slot_mapping length != number of sequences")
    _need((q_weight is None) == (k_weight is None), "q_weight and k_weight go together")
    if q_weight is not None:
        _gpu_bf16(q_weight, "q_weight")
        _gpu_bf16(k_weight, "k_weight")
        _need(q_weight.numel() == d and k_weight.numel() == d, "norm weights must be [head_dim]")
    if out is None:
        out = torch.empty((B, hq, d), dtype=_BF16, device=q.device)
    else:
        _gpu_bf16(out, "out")
        _rows_ok(out, "out")
        _need(out.shape == q.shape, "out shape != q shape")
    dll = _lib.lib()
    need = dll.nv_attn_decode_workspace_bytes_chunked(B, hq, hkv, d, int(max_seqlen_k), int(chunk_keys))
    ws, flags = _decode_workspace(q.device, need, B * hkv)
    rc = dll.nv_attn_decode_paged_fused_chunked(
        q.data_ptr(), k.data_ptr(), v.data_ptr(),
        q_weight.data_ptr() if q_weight is not None else None,
        k_weight.data_ptr() if k_weight is not None else None,
        cos_sin.data_ptr(), positions.data_ptr(), slot_mapping.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
        out.data_ptr(), cu_seqlens_k.data_ptr(), block_tables.data_ptr(), B, block_tables.shape[1],
        int(max_seqlen_k), hq, hkv, d, nb, bs, cos_sin.shape[0], float(eps), float(scale),
        q.stride(0), k.stride(0), v.stride(0), out.stride(0), ws.data_ptr(), ws.numel(), int(chunk_keys), flags,
        _stream(q))
    _lib.check(rc, "nv_attn_decode_paged_fused_chunked")
    return out


def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    """y = bf16(silu(f32(gate)) * f32(up)) for x = [gate | up] along the last dimension."""
    _need(x.is_cuda, "x must be a GPU tensor (this path has no CPU implementation)")
    _need(x.dtype == _BF16, f"x must be bfloat16, got {x.dtype}")
    _need(x.dim() >= 1 and x.shape[-1] % 16 == 0, "last dimension must be a multiple of 16")
    inter = x.shape[-1] // 2
    x2 = x.reshape(-1, x.shape[-1])
    _need(x2.stride(1) == 1 and x2.stride(0) % 8 == 0, "rows must be contiguous with a stride that is a multiple of 8")
    y = torch.empty(x.shape[:-1] + (inter,), dtype=_BF16, device=x.device)
    rc = _lib.lib().nv_silu_and_mul(x2.data_ptr(), y.data_ptr(), x2.shape[0], inter, x2.stride(0), _stream(x))
    _lib.check(rc, "nv_silu_and_mul")
    return y


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """y = bf16(f32(x) * rsqrt(mean(x^2) + eps) * f32(w)) over the last dimension."""
    _gpu_bf16(x, "x")
    _gpu_bf16(weight, "weight")
    cols = x.shape[-1]
    _need(weight.numel() == cols and weight.is_contiguous(), "weight must be [hidden] contiguous")
    x2 = x.reshape(-1, cols)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    y = torch.empty(x.shape, dtype=_BF16, device=x.device)
    rc = _lib.lib().nv_rmsnorm(x2.data_ptr(), weight.data_ptr(), y.data_ptr(), x2.shape[0], cols,
                               x2.stride(0) if x2.shape[0] > 1 else cols, float(eps), _stream(x))
    _lib.check(rc, "nv_rmsnorm")
    return y


def add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
    """(y, new_residual): new_residual = bf16(x + residual); y = rmsnorm of the unrounded sum."""
    _gpu_bf16(x, "x")
    _gpu_bf16(residual, "residual")
    _gpu_bf16(weight, "weight")
    cols = x.shape[-1]
    _need(x.shape == residual.shape, "x and residual shapes differ")
    _need(weight.numel() == cols and weight.is_contiguous(), "weight must be [hidden] contiguous")
    x2 = x.contiguous().view(-1, cols)
    r2 = residual.contiguous().view(-1, cols)
    y = torch.empty(x.shape, dtype=_BF16, device=x.device)
    new_res = torch.empty(x.shape, dtype=_BF16, device=x.device)
    rc = _lib.lib().nv_add_rmsnorm(x2.data_ptr(), r2.data_ptr(), weight.data_ptr(), y.data_ptr(),
                                   new_res.data_ptr(), x2.shape[0], cols, float(eps), _stream(x))
    _lib.check(rc, "nv_add_rmsnorm")
    return y, new_res


def _rope_common(positions, q, k, cos_sin):
    _gpu_bf16(q, "query")
    _gpu_bf16(k, "key")
    _rows_ok(q, "query")
    _rows_ok(k, "key")
    _need(positions.is_cuda and positions.dtype == torch.int64 and positions.is_contiguous()
          and positions.numel() == q.shape[0] == k.shape[0], "positions must be a contiguous int64 [tokens] GPU tensor")
    _need(cos_sin.is_cuda and cos_sin.dtype == torch.float32 and cos_sin.is_contiguous()
          and cos_sin.shape[-1] == q.shape[2], "cos_sin_cache must be contiguous fp32 [max_pos, (1,) head_dim]")


def rope(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, cos_sin: torch.Tensor):
    """NeoX rotary embedding of q [T,H_q,D] and k [T,H_kv,D]; returns new tensors."""
    _rope_common(positions, q, k, cos_sin)
    T, hq, d = q.shape
    hkv = k.shape[1]
    q_out = torch.empty((T, hq, d), dtype=_BF16, device=q.device)
    k_out = torch.empty((T, hkv, d), dtype=_BF16, device=q.device)
    rc = _lib.lib().nv_rope(q.data_ptr(), k.data_ptr(), q_out.data_ptr(), k_out.data_ptr(),
                            cos_sin.data_ptr(), positions.data_ptr(), T, hq, hkv, d, cos_sin.shape[0],
                            q.stride(0), k.stride(0), _stream(q))
    _lib.check(rc, "nv_rope")
    return q_out, k_out


def qknorm_rope_store(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: Optional[torch.Tensor],
                      q_weight: Optional[torch.Tensor], k_weight: Optional[torch.Tensor],
                      cos_sin: torch.Tensor, eps: float, slot_mapping: Optional[torch.Tensor] = None,
                      k_cache: Optional[torch.Tensor] = None, v_cache: Optional[torch.Tensor] = None,
                      want_k: bool = True):
    """Fused per-head RMSNorm (optional) + rotary + KV scatter (optional).  Returns (q_out, k_out)."""
    _rope_common(positions, q, k, cos_sin)
    T, hq, d = q.shape
    hkv = k.shape[1]
    store = slot_mapping is not None
    nb = bs = 0
    if store:
        _need(v is not None and k_cache is not None and v_cache is not None, "scatter needs v, k_cache, v_cache")
        _gpu_bf16(v, "v")
        _rows_ok(v, "v")
        _i32(slot_mapping, "slot_mapping")
        _need(k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape == v_cache.shape
              and k_cache.shape[2:] == (hkv, d), "caches must be contiguous [num_blocks, block_size, H_kv, D]")
        nb, bs = k_cache.shape[0], k_cache.shape[1]
    if q_weight is not None:
        _gpu_bf16(q_weight, "q_weight")
        _gpu_bf16(k_weight, "k_weight")
        _need(q_weight.numel() == d and k_weight.numel() == d, "norm weights must be [head_dim]")
    q_out = torch.empty((T, hq, d), dtype=_BF16, device=q.device)
    k_out = torch.empty((T, hkv, d), dtype=_BF16, device=q.device) if (want_k or not store) else None
    rc = _lib.lib().nv_qknorm_rope_store(
        q.data_ptr(), k.data_ptr(), v.data_ptr() if store else None,
        q_weight.data_ptr() if q_weight is not None else None,
        k_weight.data_ptr() if k_weight is not None else None,
        cos_sin.data_ptr(), positions.data_ptr(),
        slot_mapping.data_ptr() if store else None,
        k_cache.data_ptr() if store else None, v_cache.data_ptr() if store else None,
        q_out.data_ptr(), k_out.data_ptr() if k_out is not None else None,
        T, hq, hkv, d, cos_sin.shape[0], float(eps), nb, bs,
        q.stride(0), k.stride(0), v.stride(0) if store else 0, _stream(q))
    _lib.check(rc, "nv_qknorm_rope_store")
    return q_out, k_out
