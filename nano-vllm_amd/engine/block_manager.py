"""Paged KV-cache bookkeeping: free queue, reference counts, prefix sharing by chained block hash.

Restates the intended behaviour of the reference's nanovllm/engine/block_manager.py:37-329:
  * blocks are handed out from the head of a FIFO free queue and returned to its tail (:129-181);
    a cached block that nobody uses stays in the queue and keeps its hash until it is handed out
    again (eviction = reuse, :209-216);
  * a full block is identified by xxh64(prev_hash little-endian 8 bytes || int32 token ids)
    (:75-90), and a new request reuses the longest run of cached full blocks of its prompt (:92-127).
The reference's defects are not reproduced: `can_allocate` compares integers (the committed code
calls len() on an int, :317); a sequence's block list exists once (`seq.block_table` is that list;
the committed code appends every grown block twice, :286-287); every full block is hashed as soon
as all of its tokens are scheduled, including blocks added while a long prompt is chunked (the
committed code only hashes blocks of the first chunk, :233-247); and a prompt is never taken
entirely from the cache, so that at least one token is computed and a logit row exists.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import xxhash

from .sequence import Sequence


class KVCacheBlock:
    __slots__ = ("id", "hash", "ref_count")

    def __init__(self, block_id: int):
        self.id = block_id
        self.hash = -1        # set when the block is full and published for sharing
        self.ref_count = 0


class BlockManager:
    def __init__(self, num_blocks: int, block_size: int):
        self.num_blocks = num_blocks
        self.block_size = block_size
        self.block_pool = [KVCacheBlock(i) for i in range(num_blocks)]
        self._free: "OrderedDict[int, None]" = OrderedDict((i, None) for i in range(num_blocks))
        self.cache_blocks: dict[int, int] = {}          # hash -> block id
        self._hashed: dict[str, int] = {}               # seq id -> number of leading blocks already hashed
        self._tables: dict[str, list[int]] = {}         # seq id -> the sequence's block_table list

    # ---- hashing ---------------------------------------------------------------------
    @classmethod
    def compute_hash(cls, token_ids, prev_hash: int = -1) -> int:
        h = xxhash.xxh64()
        if prev_hash != -1:
            h.update(int(prev_hash).to_bytes(8, "little"))
        h.update(np.asarray(token_ids, dtype=np.int32).tobytes())
        return h.intdigest()

    @property
    def num_free_blocks(self) -> int:
        return len(self._free)

    def can_allocate(self, num_blocks: int) -> bool:
        return len(self._free) >= num_blocks

    # ---- prefix lookup ---------------------------------------------------------------
    def get_computed_blocks(self, token_ids) -> tuple[list[int], int]:
        """Longest cached prefix of full blocks, never the whole prompt."""
        bs = self.block_size
        usable = (len(token_ids) - 1) // bs          # leaves >= 1 token to compute
        blocks, prev = [], -1
        for i in range(usable):
            prev = self.compute_hash(token_ids[i * bs:(i + 1) * bs], prev)
            block_id = self.cache_blocks.get(prev)
            if block_id is None:
                break
            blocks.append(block_id)
        return blocks, len(blocks) * bs

    # ---- allocation ------------------------------------------------------------------
    def _take_free_block(self) -> int:
        if not self._free:
            raise ValueError("No free blocks available")
        block_id, _ = self._free.popitem(last=False)
        block = self.block_pool[block_id]
        if block.hash != -1:                         # evict a cached-but-unused block
            if self.cache_blocks.get(block.hash) == block_id:
                del self.cache_blocks[block.hash]
            block.hash = -1
        block.ref_count = 1
        return block_id

    def allocate_slots(self, seq: Sequence, cached_blocks: list[int], num_new_blocks: int) -> list[int]:
        """Admission: share `cached_blocks`, add `num_new_blocks` fresh ones; returns the block table
        (the same list object is installed as seq.block_table)."""
        for block_id in cached_blocks:
            block = self.block_pool[block_id]
            if block.ref_count == 0:
                del self._free[block_id]             # it was idle in the free queue
            block.ref_count += 1
        table = list(cached_blocks) + [self._take_free_block() for _ in range(num_new_blocks)]
        seq.block_table = table
        self._tables[seq.id] = table
        self._hashed[seq.id] = len(cached_blocks)
        return table

    def append_slots(self, seq: Sequence, num_new_tokens: int) -> None:
        """Grow the table so that num_computed + num_new_tokens tokens fit."""
        table = self._tables[seq.id]
        needed = -(-(seq.num_computed_tokens + num_new_tokens) // self.block_size) - len(table)
        if needed > 0:
            if not self.can_allocate(needed):
                raise ValueError(f"Cannot allocate {needed} blocks for seq {seq.id}")
            table.extend(self._take_free_block() for _ in range(needed))

    def publish_full_blocks(self, seq: Sequence, num_tokens_in_cache: int) -> None:
        """Hash and publish every block of `seq` that is completely covered by the first
        `num_tokens_in_cache` tokens (those already in the cache plus those scheduled now)."""
        bs = self.block_size
        table = self._tables[seq.id]
        done = self._hashed[seq.id]
        num_tokens_in_cache = min(num_tokens_in_cache, len(seq) - seq.num_pending)   # only known token values
        full = min(num_tokens_in_cache // bs, len(table))
        if full <= done:
            return
        prev = self.block_pool[table[done - 1]].hash if done else -1
        tokens = seq.token_ids
        for i in range(done, full):
            prev = self.compute_hash(tokens[i * bs:(i + 1) * bs], prev)
            block = self.block_pool[table[i]]
            block.hash = prev
            self.cache_blocks.setdefault(prev, block.id)
        self._hashed[seq.id] = full

    def free_sequence(self, seq_id: str) -> None:
        table = self._tables.pop(seq_id, None)
        self._hashed.pop(seq_id, None)
        if table is None:
            return
        for block_id in table:
            block = self.block_pool[block_id]
            block.ref_count -= 1
            if block.ref_count == 0:
                self._free[block_id] = None          # tail of the queue; keeps its hash until reused
