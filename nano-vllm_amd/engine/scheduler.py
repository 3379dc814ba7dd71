"""Continuous-batching scheduler with chunked prefill (host Python).

Policy of the reference's nanovllm/engine/scheduler.py:30-160: running sequences first (a decode
token, or the next prompt chunk of at most `long_prefill_token_threshold` tokens), then waiting
sequences while the token budget `max_num_batched_tokens` lasts; admission looks up the prefix
cache and allocates blocks for the first chunk only; a sequence finishes on `max_tokens` or EOS and
its blocks are released.  `max_num_seqs` is enforced only when set (the reference stores it and never
reads it).

When a running sequence cannot get a block the reference prints and drops it from the queue
(scheduler.py:52-57), which loses the request.  Here the youngest running sequence is preempted
instead: its blocks are released, it returns to the head of the waiting queue and its K/V
(prompt and the tokens generated so far) is recomputed on re-admission, through the prefix cache
where its blocks survived.  Admission also asks for enough free blocks for the whole prompt (a
check, not a reservation), so that a long prompt is not admitted only to be preempted again.
"""
from __future__ import annotations

from collections import deque

from .block_manager import BlockManager
from .sequence import PENDING, Sequence, SequenceStatus


class Scheduler:
    def __init__(self, num_kvcache_blocks: int, kvcache_block_size: int = 256, max_num_batched_tokens: int = 4000,
                 long_prefill_token_threshold: int = 512, eos: int | None = None, max_num_seqs: int | None = None):
        self.block_size = kvcache_block_size
        self.max_scheduled_tokens = max_num_batched_tokens
        self.long_prefill_token_threshold = long_prefill_token_threshold
        self.eos = eos
        self.max_num_seqs = max_num_seqs
        self.block_manager = BlockManager(num_kvcache_blocks, kvcache_block_size)
        self.waiting: deque[Sequence] = deque()
        self.running: deque[Sequence] = deque()

    def add(self, seq: Sequence) -> None:
        seq.status = SequenceStatus.WAITING
        self.waiting.append(seq)

    def is_finished(self) -> bool:
        return not self.waiting and not self.running

    def schedule(self) -> tuple[list[Sequence], bool]:
        """Returns (scheduled sequences in batch order, whether any new sequence was admitted)."""
        bm = self.block_manager
        budget = self.max_scheduled_tokens
        scheduled: list[Sequence] = []
        # 1. running sequences, oldest first: decode token, next prompt chunk, or next chunk of a
        #    recomputation.  Tokens [computed, computed + n_new) of prompt + outputs are processed.
        running = list(self.running)
        preempted: list[Sequence] = []
        i = 0
        while i < len(running):
            seq = running[i]
            seq.num_scheduled_tokens = 0
            if budget <= 0 or seq.is_finished:
                i += 1
                continue
            # a decode row has exactly one token left; a prompt (or a recomputation) is chunked only when
            # the threshold is positive, as on admission
            n_new = len(seq) - seq.num_computed_tokens
            if 0 < self.long_prefill_token_threshold < n_new:
                n_new = self.long_prefill_token_threshold
            n_new = min(n_new, budget)
            got = False
            while True:
                try:
                    bm.append_slots(seq, n_new)
                    got = True
                    break
                except ValueError:
                    victim = self._pick_victim(running, i)
                    if victim is None:
                        break                  # nobody can give way right now: sits this step out
                    running.remove(victim)
                    self._preempt(victim)
                    preempted.append(victim)
                    if victim is seq:
                        break
            if got:
                seq.num_scheduled_tokens = n_new
                bm.publish_full_blocks(seq, seq.num_computed_tokens + n_new)
                budget -= n_new
                scheduled.append(seq)
                i += 1
            elif i < len(running) and running[i] is seq:
                i += 1                         # still running, not scheduled
        if preempted:
            self.running = deque(running)
            for victim in preempted:           # youngest was preempted first: oldest ends up at the head
                self.waiting.appendleft(victim)
        # 2. admit waiting sequences (not in a step that had to preempt: the cache is full)
        admitted = 0
        while self.waiting and budget > 0 and not preempted:
            if self.max_num_seqs is not None and len(self.running) >= self.max_num_seqs:
                break
            seq = self.waiting[0]
            tokens = seq.token_ids if seq.output_tokens else seq.prompt_tokens   # all of it after a preemption
            cached_blocks, n_cached = bm.get_computed_blocks(tokens)
            n_new = len(tokens) - n_cached
            if 0 < self.long_prefill_token_threshold < n_new:
                n_new = self.long_prefill_token_threshold
            n_new = min(n_new, budget)
            # blocks for tokens [n_cached, n_cached + n_new); n_cached is block aligned
            blocks_needed = -(-n_new // self.block_size)
            blocks_whole = -(-(len(tokens) + 1) // self.block_size) - len(cached_blocks)   # prompt + first new token
            idle_hits = sum(1 for b in cached_blocks if bm.block_pool[b].ref_count == 0)
            if not bm.can_allocate(max(blocks_needed, blocks_whole) + idle_hits):
                break
            bm.allocate_slots(seq, cached_blocks, blocks_needed)
            seq.num_computed_tokens = n_cached
            seq.num_cached_tokens = n_cached
            seq.num_scheduled_tokens = n_new
            seq.status = SequenceStatus.RUNNING
            bm.publish_full_blocks(seq, n_cached + n_new)
            budget -= n_new
            self.waiting.popleft()
            self.running.append(seq)
            if seq.num_preemptions and len(self.running) > 1 and self.running[-2].arrival > seq.arrival:
                # a re-admitted sequence takes its place by arrival again: otherwise the oldest request
                # would sit at the tail and be the first victim of the next preemption
                self.running = deque(sorted(self.running, key=lambda s: s.arrival))
            scheduled.append(seq)
            admitted += 1
        return scheduled, admitted > 0

    @staticmethod
    def _pick_victim(running: list[Sequence], current: int) -> Sequence | None:
        """Youngest running sequence (by arrival: `running` is kept in arrival order) at or after position
        `current` whose token values are all known on the host (a sequence with a token still in flight
        cannot be recomputed yet)."""
        for j in range(len(running) - 1, current - 1, -1):
            cand = running[j]
            if not cand.is_finished and cand.num_pending == 0:
                return cand
        return None

    def _preempt(self, seq: Sequence) -> None:
        """Recompute-preemption: release the blocks, forget the computed count, back to waiting."""
        self.block_manager.free_sequence(seq.id)
        seq.block_table = []
        seq.num_computed_tokens = 0
        seq.num_scheduled_tokens = 0
        seq.num_preemptions += 1
        seq.status = SequenceStatus.WAITING

    def _retire(self, finished: list[Sequence]) -> None:
        for seq in finished:
            self.block_manager.free_sequence(seq.id)
            seq.block_table = []
        self.running = deque(s for s in self.running if not s.is_finished)

    def advance_pending(self, seqs: list[Sequence]) -> list[Sequence]:
        """Pipelined decode: account a launched decode-only step before its tokens are known.  Every
        sequence gets a PENDING output token; those that reach max_tokens finish now (their last value
        arrives with resolve_pending).  Returns the sequences that finished."""
        finished = []
        for seq in seqs:
            seq.num_computed_tokens += seq.num_scheduled_tokens
            seq.num_scheduled_tokens = 0
            seq.add_token(PENDING)
            seq.num_pending += 1
            if len(seq.output_tokens) >= seq.sampling_params.max_tokens:
                seq.status = SequenceStatus.FINISHED
                finished.append(seq)
        if finished:
            self._retire(finished)
        return finished

    def resolve_pending(self, seqs: list[Sequence], token_ids: list[int]) -> list[Sequence]:
        """Fill in the values of a step accounted by advance_pending.  A sequence that turns out to
        have sampled EOS finishes here; the token of a step that was launched for it in the meantime
        is dropped when that step resolves.  Returns the sequences that finished on EOS."""
        finished = []
        for seq, token_id in zip(seqs, token_ids):
            if seq.stopped_early:                 # launched before its EOS was seen: discard
                if seq.num_pending:
                    seq.output_tokens.pop()
                    seq.num_pending -= 1
                continue
            assert seq.num_pending >= 1, "resolve_pending without a pending token"
            seq.output_tokens[len(seq.output_tokens) - seq.num_pending] = token_id
            seq.num_pending -= 1
            if token_id == self.eos and not seq.is_finished:
                seq.status = SequenceStatus.FINISHED
                seq.stopped_early = True
                finished.append(seq)
        if finished:
            self._retire(finished)
        return finished

    def update_from_output(self, seqs: list[Sequence], token_ids: list[int | None]) -> list[Sequence]:
        """Account the step: advance computed counts, append sampled tokens (None = prompt not yet
        complete), finish and free.  Returns the sequences that finished."""
        finished = []
        for seq, token_id in zip(seqs, token_ids):
            seq.num_computed_tokens += seq.num_scheduled_tokens
            seq.num_scheduled_tokens = 0
            if token_id is not None:
                seq.add_token(token_id)
                if len(seq.output_tokens) >= seq.sampling_params.max_tokens or token_id == self.eos:
                    seq.status = SequenceStatus.FINISHED
                    finished.append(seq)
        if finished:
            self._retire(finished)
        return finished
