"""Qwen3-shaped decoder used as the caller harness of the attention hot path.

Same module tree and HF parameter names as the reference's nanovllm/model/qwen3.py:10-229
(`model.layers.N.self_attn.{q,k,v,o}_proj`, `.q_norm`, `.k_norm`, `input_layernorm`,
`post_attention_layernorm`, `model.norm`, `lm_head`, tied embeddings), same pre-norm residual
threading (:148-172).  The GEMMs, the embedding and SwiGLU stay plain torch ops (out of scope of
the hand-written path).  Two differences in wiring:
  * the attention block can run the chain q_norm -> k_norm -> rope -> store_kvcache as ONE HIP
    launch (`nv_qknorm_rope_store`) followed by `Attention.attend`; `fused=False` keeps the
    reference's module-by-module order (:62-75);
  * `layers` selects where Attention / RMSNorm / get_rope come from.  The default is this package's
    HIP-backed modules; tests pass a module with CPU stand-ins to exercise the host plumbing.
Models without q/k-norm (Llama-style) set `qk_norm=False`.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..parallel import HeadGather

from ..utils.cache_context import get_context


@dataclass
class ModelConfig:
    vocab_size: int = 151936
    hidden_size: int = 1024
    intermediate_size: int = 3072
    num_hidden_layers: int = 28
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    max_position_embeddings: int = 40960
    tie_word_embeddings: bool = True
    attention_bias: bool = False
    qk_norm: bool = True
    rope_scaling: dict | None = None
    eos_token_id: int | None = None

    @classmethod
    def qwen3_0_6b(cls, **kw):
        return cls(**kw)

    @classmethod
    def qwen3_8b(cls, **kw):
        return cls(hidden_size=4096, intermediate_size=12288, num_hidden_layers=36, num_attention_heads=32,
                   tie_word_embeddings=False, **kw)

    @classmethod
    def qwen3_14b(cls, **kw):
        return cls(hidden_size=5120, intermediate_size=17408, num_hidden_layers=40, num_attention_heads=40,
                   tie_word_embeddings=False, **kw)

    @classmethod
    def qwen3_32b(cls, **kw):
        return cls(hidden_size=5120, intermediate_size=25600, num_hidden_layers=64, num_attention_heads=64,
                   tie_word_embeddings=False, **kw)

    @classmethod
    def llama31_8b(cls, **kw):
        return cls(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                   num_attention_heads=32, rope_theta=5e5, max_position_embeddings=131072, rms_norm_eps=1e-5,
                   tie_word_embeddings=False, qk_norm=False,
                   rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                     original_max_position_embeddings=8192), **kw)


def _default_layers():
    from .. import layers as _pkg  # noqa: F401
    from ..layers import activation, attention, layernorm, rope

    class _L:
        SiluAndMul = activation.SiluAndMul
        Attention = attention.Attention
        RMSNorm = layernorm.RMSNorm
        make_rope = staticmethod(rope.make_rope)
        fused_qknorm_rope_store = staticmethod(_hip_fused)
        fused_decode_attention = staticmethod(_hip_fused_decode)
    return _L


def _hip_fused(positions, q, k, v, q_w, k_w, rope, eps, attn):
    """q_norm + k_norm + rope + KV scatter in one launch; returns the rotated q."""
    from .. import ops
    ctx = get_context()
    q_out, _ = ops.qknorm_rope_store(positions, q, k, v, q_w, k_w, rope.cos_sin_cache, eps, ctx.slot_mapping,
                                     attn.k_cache, attn.v_cache, want_k=False)
    return q_out


def _hip_fused_decode(positions, q, k, v, q_w, k_w, rope, eps, attn):
    """A pure decode step (every sequence has one new token, token b = sequence b): q/k-norm, rotary, KV
    scatter and paged attention in ONE launch.  None when the step or the head shape is not one the fused
    launch takes; the caller then runs the two launches."""
    from .. import ops
    ctx = get_context()
    if ctx.max_seqlen_q > 1 or q.shape[0] != ctx.cu_seqlens_k.numel() - 1:
        return None
    if not ops.decode_fused_supported(q.shape[1], k.shape[1], attn.k_cache.shape[1]):
        return None
    return ops.attn_decode_paged_fused(positions, q, k, v, q_w, k_w, rope.cos_sin_cache, eps, ctx.slot_mapping,
                                       attn.k_cache, attn.v_cache, ctx.cu_seqlens_k, ctx.block_tables,
                                       ctx.max_seqlen_k, attn.attn_scale,
                                       chunk_keys=getattr(ctx, "decode_chunk_keys", 0) or 0)


class Qwen3Attention(nn.Module):
    """With `shard` (a parallel.HeadShard) the module owns only this rank's kv heads and their query
    heads: q/k/v_proj hold the matching rows, the KV cache has H_kv/N heads, and the head outputs
    of all ranks are all-gathered before the (replicated) o_proj."""

    def __init__(self, cfg: ModelConfig, L, fused: bool, shard=None, group=None, gather=None):
        super().__init__()
        self.shard, self.group = shard, group
        # one HeadGather (persistent all-gather buffers) is shared by all layers of a model
        self._gather = gather if gather is not None else (HeadGather(shard, group) if shard is not None else None)
        self.head_dim = cfg.head_dim
        self.num_heads = shard.local_heads if shard else cfg.num_attention_heads
        self.num_kv_heads = shard.local_kv_heads if shard else cfg.num_key_value_heads
        self.q_size, self.kv_size = self.num_heads * self.head_dim, self.num_kv_heads * self.head_dim
        self.eps = cfg.rms_norm_eps
        self.q_proj = nn.Linear(cfg.hidden_size, self.q_size, bias=cfg.attention_bias)
        self.k_proj = nn.Linear(cfg.hidden_size, self.kv_size, bias=cfg.attention_bias)
        self.v_proj = nn.Linear(cfg.hidden_size, self.kv_size, bias=cfg.attention_bias)
        self.o_proj = nn.Linear(cfg.num_attention_heads * self.head_dim, cfg.hidden_size, bias=cfg.attention_bias)
        self.rope = L.make_rope(self.head_dim, cfg.max_position_embeddings, cfg.rope_theta, cfg.rope_scaling)
        self.qk_norm = cfg.qk_norm
        if cfg.qk_norm:
            self.q_norm = L.RMSNorm(self.head_dim, cfg.rms_norm_eps)
            self.k_norm = L.RMSNorm(self.head_dim, cfg.rms_norm_eps)
        self.attention = L.Attention(self.num_heads, self.head_dim, self.num_kv_heads, self.head_dim ** -0.5,
                                     cfg.max_position_embeddings)
        self._fused = L.fused_qknorm_rope_store if fused else None
        self._fused_decode = getattr(L, "fused_decode_attention", None) if fused else None

    def fuse_weights(self) -> None:
        """Make q/k/v_proj.weight views of ONE [q+2kv, hidden] tensor so the three projections run
        as a single GEMM; parameter names and values are unchanged (call after loading weights)."""
        if self.q_proj.bias is not None:
            return
        w = torch.cat([self.q_proj.weight.data, self.k_proj.weight.data, self.v_proj.weight.data], dim=0)
        self.q_proj.weight.data = w[:self.q_size]
        self.k_proj.weight.data = w[self.q_size:self.q_size + self.kv_size]
        self.v_proj.weight.data = w[self.q_size + self.kv_size:]
        self._qkv_weight = w

    def _project(self, x: torch.Tensor):
        w = getattr(self, "_qkv_weight", None)
        if w is not None and w.device == x.device and w.dtype == x.dtype:
            qkv = F.linear(x, w)                       # [T, q+2kv]; the slices below keep its token stride
            q = qkv[:, :self.q_size]
            k = qkv[:, self.q_size:self.q_size + self.kv_size]
            v = qkv[:, self.q_size + self.kv_size:]
        else:
            q, k, v = self.q_proj(x), self.k_proj(x), self.v_proj(x)
        T = x.shape[0]
        return (q.unflatten(1, (self.num_heads, self.head_dim)), k.unflatten(1, (self.num_kv_heads, self.head_dim)),
                v.unflatten(1, (self.num_kv_heads, self.head_dim)))

    def forward(self, x: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        q, k, v = self._project(x)
        if self._fused is not None:
            q_w = self.q_norm.weight if self.qk_norm else None
            k_w = self.k_norm.weight if self.qk_norm else None
            o = None
            if self._fused_decode is not None:
                o = self._fused_decode(positions, q, k, v, q_w, k_w, self.rope, self.eps, self.attention)
            if o is None:
                q = self._fused(positions, q, k, v, q_w, k_w, self.rope, self.eps, self.attention)
                o = self.attention.attend(q)
        else:
            if self.qk_norm:
                q, k = self.q_norm(q), self.k_norm(k)
            q, k = self.rope(positions, q, k)
            o = self.attention(q, k, v)
        if self.shard is not None and self.shard.world > 1:
            o = self._gather(o.reshape(-1, self.num_heads, self.head_dim))
        return self.o_proj(o.reshape(o.shape[0], -1))


class Qwen3MLP(nn.Module):
    def __init__(self, cfg: ModelConfig, L):
        super().__init__()
        self.act_fn = L.SiluAndMul()
        self.gate_proj = nn.Linear(cfg.hidden_size, cfg.intermediate_size, bias=False)
        self.up_proj = nn.Linear(cfg.hidden_size, cfg.intermediate_size, bias=False)
        self.down_proj = nn.Linear(cfg.intermediate_size, cfg.hidden_size, bias=False)

    def fuse_weights(self) -> None:
        """gate_proj / up_proj weights become views of one [2I, hidden] tensor: one GEMM for both."""
        w = torch.cat([self.gate_proj.weight.data, self.up_proj.weight.data], dim=0)
        n = self.gate_proj.weight.shape[0]
        self.gate_proj.weight.data = w[:n]
        self.up_proj.weight.data = w[n:]
        self._gate_up_weight = w

    def forward(self, x):
        w = getattr(self, "_gate_up_weight", None)
        if w is not None and w.device == x.device and w.dtype == x.dtype:
            return self.down_proj(self.act_fn(F.linear(x, w)))
        return self.down_proj(self.act_fn(torch.cat([self.gate_proj(x), self.up_proj(x)], dim=-1)))


class Qwen3DecoderLayer(nn.Module):
    def __init__(self, cfg: ModelConfig, L, fused: bool, shard=None, group=None, gather=None):
        super().__init__()
        self.self_attn = Qwen3Attention(cfg, L, fused, shard, group, gather)
        self.mlp = Qwen3MLP(cfg, L)
        self.input_layernorm = L.RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)
        self.post_attention_layernorm = L.RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)

    def forward(self, positions, hidden, residual):
        if residual is None:
            residual = hidden
            hidden = self.input_layernorm(hidden)
        else:
            hidden, residual = self.input_layernorm(hidden, residual)
        hidden = self.self_attn(hidden, positions)
        hidden, residual = self.post_attention_layernorm(hidden, residual)
        return self.mlp(hidden), residual


class Qwen3Model(nn.Module):
    def __init__(self, cfg: ModelConfig, L, fused: bool, shard=None, group=None):
        super().__init__()
        self.embed_tokens = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        gather = HeadGather(shard, group) if shard is not None else None
        self.layers = nn.ModuleList(Qwen3DecoderLayer(cfg, L, fused, shard, group, gather)
                                    for _ in range(cfg.num_hidden_layers))
        self.norm = L.RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)

    def forward(self, input_ids, positions):
        hidden, residual = self.embed_tokens(input_ids), None
        ctx = get_context()
        for idx, layer in enumerate(self.layers):
            ctx.layer_idx = idx
            hidden, residual = layer(positions, hidden, residual)
        hidden, _ = self.norm(hidden, residual)
        return hidden


class Qwen3ForCausalLM(nn.Module):
    def __init__(self, cfg: ModelConfig, layers=None, fused: bool = True, shard=None, group=None):
        """shard: optional parallel.HeadShard of this rank (attention sharded by kv head over the
        ranks of `group`, everything else replicated)."""
        super().__init__()
        L = layers if layers is not None else _default_layers()
        self.config, self.shard, self.group = cfg, shard, group
        if shard is not None and (shard.num_heads, shard.num_kv_heads) != (cfg.num_attention_heads,
                                                                             cfg.num_key_value_heads):
            raise ValueError("HeadShard head counts do not match the model config")
        self.model = Qwen3Model(cfg, L, fused and hasattr(L, "fused_qknorm_rope_store"), shard, group)
        self.lm_head = nn.Linear(cfg.hidden_size, cfg.vocab_size, bias=False)
        if cfg.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight

    def load_full_state_dict(self, full: dict) -> None:
        """Load an unsharded state dict: with a shard, q/k/v_proj take the rows of this rank's heads."""
        sh, d = self.shard, self.config.head_dim
        sd = {}
        for name, t in full.items():
            if sh is not None and ".self_attn." in name:
                part = name.rsplit(".", 2)[-2]
                if part == "q_proj":
                    t = t[sh.q_heads.start * d:sh.q_heads.stop * d]
                elif part in ("k_proj", "v_proj"):
                    t = t[sh.kv_heads.start * d:sh.kv_heads.stop * d]
            sd[name] = t
        self.load_state_dict(sd)

    def fuse_weights(self) -> "Qwen3ForCausalLM":
        """Fuse the q/k/v and gate/up projection weights of every layer (after weights are loaded and
        the model sits on its final device / dtype).  Outputs are unchanged: the same rows of the same
        matrices are multiplied, in one GEMM instead of three / two."""
        for layer in self.model.layers:
            layer.self_attn.fuse_weights()
            layer.mlp.fuse_weights()
        return self

    def forward(self, input_ids, positions, logits_rows=None):
        """logits_rows: optional index tensor; when given only those token rows go through lm_head."""
        hidden = self.model(input_ids, positions)
        if logits_rows is not None:
            hidden = hidden[logits_rows]
        return self.lm_head(hidden)
