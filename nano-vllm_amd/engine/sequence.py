"""Per-request host state (plain Python; mirrors the fields of the reference's
nanovllm/engine/sequence.py:11-72 that the scheduler, block manager and runner read)."""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum
from itertools import count


@dataclass
class SamplingParams:
    """nanovllm/sampling_params.py:3-6.  temperature == 0 selects argmax decoding (the reference's
    sampler rejects it, nanovllm/utils/sampler.py:20-22)."""
    temperature: float = 1.0
    max_tokens: int = 256


class SequenceStatus(Enum):
    WAITING = "waiting"
    RUNNING = "running"
    FINISHED = "finished"


PENDING = -1   # placeholder for a sampled token whose value is still on the device


class Sequence:
    _ids = count()

    def __init__(self, prompt_tokens: list[int], sampling_params: SamplingParams | None = None):
        self.arrival = next(Sequence._ids)    # arrival order: the scheduler keeps `running` sorted by it
        self.id = f"seq_{self.arrival}"
        self.status = SequenceStatus.WAITING
        self.prompt_tokens = list(prompt_tokens)
        self.output_tokens: list[int] = []
        self.num_computed_tokens = 0      # tokens whose K/V is in the cache
        self.num_cached_tokens = 0        # of those, prompt tokens found in the prefix cache at admission
        self.num_scheduled_tokens = 0     # tokens to process in the coming step
        self.block_table: list[int] = []  # physical block ids, logical order
        self.num_pending = 0              # trailing output tokens that are still PENDING (0 or 1)
        self.stopped_early = False        # finished on EOS while a later step was already in flight
        self.num_preemptions = 0          # times its blocks were taken away and its K/V recomputed
        self.sampling_params = sampling_params or SamplingParams()

    def __len__(self) -> int:
        return len(self.prompt_tokens) + len(self.output_tokens)

    num_tokens = property(__len__)

    @property
    def is_finished(self) -> bool:
        return self.status == SequenceStatus.FINISHED

    @property
    def is_prefill(self) -> bool:
        """No sampled token yet (sequence.py:45-49)."""
        return not self.output_tokens

    @property
    def completes_step(self) -> bool:
        """The scheduled tokens reach the end of the sequence, so the step's last logit row samples
        its next token.  False while a prompt is being chunked or a preempted sequence is being
        recomputed."""
        return self.num_computed_tokens + self.num_scheduled_tokens >= len(self)

    @property
    def last_token(self) -> int:
        return self.output_tokens[-1] if self.output_tokens else self.prompt_tokens[-1]

    @property
    def token_ids(self) -> list[int]:
        return self.prompt_tokens + self.output_tokens

    def token_at(self, pos: int) -> int:
        n = len(self.prompt_tokens)
        return self.prompt_tokens[pos] if pos < n else self.output_tokens[pos - n]

    def add_token(self, token_id: int) -> None:
        self.output_tokens.append(token_id)

    def __repr__(self) -> str:
        return f"Sequence({self.id}, {self.status.value}, tokens={len(self)}, blocks={len(self.block_table)})"
