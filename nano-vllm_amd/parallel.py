"""Head-wise sharding of the attention path across GPUs of one node, and the all-gather that
reassembles the head outputs.

The reference has no distributed code at all; this is new.  Every (sequence, kv head) attention
problem is independent, so rank r of N owns kv heads [r*H_kv/N, (r+1)*H_kv/N) and the G*H_kv/N
query heads that use them; its KV cache is [2, L, num_blocks, block_size, H_kv/N, D].  Scheduling
metadata (slot_mapping, block_tables, cu_seqlens, positions) is replicated.  Per-head arithmetic
is unchanged, so the gathered output equals the single-GPU output bit for bit.  The only exchange
is one all-gather of o per layer (RCCL over xGMI through torch.distributed's "nccl" backend;
"gloo" in the CPU tests): each rank contributes [T, H_q/N, D] bf16.  Independent request batches
need no communication (data-parallel replicas).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class HeadShard:
    rank: int
    world: int
    num_heads: int       # global query heads
    num_kv_heads: int    # global kv heads

    def __post_init__(self):
        if self.num_kv_heads % self.world or self.num_heads % self.num_kv_heads:
            raise ValueError(f"cannot shard {self.num_heads}/{self.num_kv_heads} heads over {self.world} ranks")

    @property
    def kv_heads(self) -> range:
        n = self.num_kv_heads // self.world
        return range(self.rank * n, (self.rank + 1) * n)

    @property
    def q_heads(self) -> range:
        g = self.num_heads // self.num_kv_heads
        return range(self.kv_heads.start * g, self.kv_heads.stop * g)

    @property
    def local_heads(self) -> int:
        return len(self.q_heads)

    @property
    def local_kv_heads(self) -> int:
        return len(self.kv_heads)

    def slice_q(self, q: torch.Tensor) -> torch.Tensor:
        """[T, H_q, D] -> this rank's [T, H_q/N, D] (a view)."""
        return q[:, self.q_heads.start:self.q_heads.stop]

    def slice_kv(self, kv: torch.Tensor) -> torch.Tensor:
        """[..., H_kv, D] -> this rank's [..., H_kv/N, D] (a view)."""
        return kv[..., self.kv_heads.start:self.kv_heads.stop, :]


def all_gather_heads(o_local: torch.Tensor, shard: HeadShard, group=None, out: torch.Tensor | None = None,
                     staging: torch.Tensor | None = None) -> torch.Tensor:
    """[T, H_q/N, D] on every rank -> [T, H_q, D] on every rank (heads in global order).
    One all-gather into a rank-major [N*T, H_q/N, D] staging buffer, then one strided copy that puts rank r's
    block at heads [r*H_q/N, (r+1)*H_q/N)."""
    T, hl, d = o_local.shape
    if shard.world == 1:
        return o_local
    if staging is None:
        staging = torch.empty((shard.world * T, hl, d), dtype=o_local.dtype, device=o_local.device)
    dist.all_gather_into_tensor(staging, o_local.contiguous(), group=group)   # rank-major concatenation
    if out is None:
        out = torch.empty((T, shard.world * hl, d), dtype=o_local.dtype, device=o_local.device)
    out.view(T, shard.world, hl, d).copy_(staging.view(shard.world, T, hl, d).permute(1, 0, 2, 3))
    return out


class HeadGather:
    """all_gather_heads with buffers that persist across calls: the model path allocates nothing per
    layer (one instance serves every layer of a model: layers run one after another on one stream),
    which also makes the step capturable in a HIP graph once the buffers have reached their size.
    The buffers grow (a new allocation) only when a step has more tokens than any step before; the
    outgrown ones stay alive for the life of the object, because decode graphs captured earlier replay
    their all-gather and copy at the old addresses (freed, that memory could be handed to another tensor
    and be overwritten on every replay).  `reserve(max_tokens)` before the first capture avoids the growth.

    Why the strided copy stays: the all-gather delivers rank-major [N, T, H_q/N, D], o_proj wants
    token-major rows [T, H_q*D].  Consuming the rank-major buffer directly would turn o_proj into N
    accumulating GEMMs of depth H_q*D/N each; at decode sizes one 4 MiB device copy is cheaper than
    N - 1 extra GEMM launches."""

    def __init__(self, shard: HeadShard, group=None):
        self.shard, self.group = shard, group
        self._staging: torch.Tensor | None = None
        self._out: torch.Tensor | None = None
        self._retired: list[torch.Tensor] = []

    def reserve(self, T: int, hl: int, d: int, dtype: torch.dtype, device) -> None:
        """Size the buffers for steps of up to T tokens (call before capturing graphs)."""
        n = self.shard.world
        device = torch.device(device)
        if (self._staging is None or self._staging.shape[0] < n * T or self._staging.device != device
                or self._staging.dtype != dtype or self._staging.shape[1:] != (hl, d)):
            if self._staging is not None:
                self._retired += [self._staging, self._out]
            self._staging = torch.empty((n * T, hl, d), dtype=dtype, device=device)
            self._out = torch.empty((T, n * hl, d), dtype=dtype, device=device)

    def _reserve(self, T: int, hl: int, d: int, like: torch.Tensor) -> None:
        self.reserve(T, hl, d, like.dtype, like.device)

    def __call__(self, o_local: torch.Tensor) -> torch.Tensor:
        if self.shard.world == 1:
            return o_local
        T, hl, d = o_local.shape
        self._reserve(T, hl, d, o_local)
        return all_gather_heads(o_local, self.shard, self.group, out=self._out[:T],
                                staging=self._staging[:self.shard.world * T])


def broadcast_tokens(tokens: torch.Tensor, group=None, src: int = 0) -> torch.Tensor:
    """Make every rank of a head-sharded engine continue with rank `src`'s sampled tokens.  Logits are
    replicated, but a random draw (temperature > 0) comes from each rank's own generator; ranks that
    disagreed on a token would schedule different batches and the next all-gather would mismatch."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(tokens, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return tokens
