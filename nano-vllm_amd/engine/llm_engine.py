"""Engine loop: add_request -> (schedule -> run -> update)* (reference: nanovllm/engine/llm_engine.py:16-175).

Works on token ids; a tokenizer is optional (there is no network here to fetch one).  Metrics
are the quantities the reference's GenerationMetrics prints (nanovllm/utils/cache_context.py:62-74):
time to first token, mean time per output token, output tokens per second.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import torch

from .model_runner import ModelRunner
from .scheduler import Scheduler
from .sequence import SamplingParams, Sequence


@dataclass
class GenerationMetrics:
    num_prompt_tokens: int = 0
    num_generated_tokens: int = 0
    wall_time: float = 0.0
    step_times: list = field(default_factory=list)          # (seconds, n_generated, decode_only)
    first_token_time: dict = field(default_factory=dict)    # seq id -> seconds since start

    def summary(self) -> dict:
        decode = [t for t, _, only in self.step_times if only]                    # steps with decode rows only
        decode_tokens = sum(n for _, n, only in self.step_times if only)
        ttft = sorted(self.first_token_time.values())
        return {
            "prompt_tokens": self.num_prompt_tokens,
            "generated_tokens": self.num_generated_tokens,
            "wall_s": self.wall_time,
            "output_tok_per_s": self.num_generated_tokens / self.wall_time if self.wall_time else 0.0,
            "decode_steps": len(decode),
            "decode_only_tok_per_s": decode_tokens / sum(decode) if decode else 0.0,
            "mean_tpot_ms": 1e3 * sum(decode) / len(decode) if decode else 0.0,
            "ttft_mean_s": sum(ttft) / len(ttft) if ttft else 0.0,
            "ttft_max_s": ttft[-1] if ttft else 0.0,
        }


def _is_decode_row(s: Sequence) -> bool:
    """One scheduled token that continues generated text (not a prompt chunk, not a recomputation)."""
    return bool(s.output_tokens) and s.num_scheduled_tokens == 1 and s.completes_step


class LLMEngine:
    def __init__(self, model: torch.nn.Module, cfg, block_size: int = 256, num_kvcache_blocks: int | None = None,
                 max_num_batched_tokens: int = 4000, long_prefill_token_threshold: int = 512,
                 gpu_memory_utilization: float = 0.9, device="cuda", eos: int | None = None,
                 max_num_seqs: int | None = None, tokenizer=None, use_graphs: bool = False,
                 pipelined_decode: bool = False):
        """pipelined_decode: decode-only steps are enqueued one ahead of the host: step t+1 is scheduled
        and launched (its input tokens are taken on the device) while step t runs, and the values of
        step t are read afterwards.  Tokens are identical to the synchronous loop; a sequence that
        samples EOS costs one surplus row in the step that was already in flight."""
        self.runner = ModelRunner(model, cfg, block_size, num_kvcache_blocks, gpu_memory_utilization, device,
                                  use_graphs=use_graphs)
        self.scheduler = Scheduler(self.runner.num_kvcache_blocks, block_size, max_num_batched_tokens,
                                   long_prefill_token_threshold, eos if eos is not None else cfg.eos_token_id,
                                   max_num_seqs)
        self.tokenizer = tokenizer
        self.pipelined_decode = pipelined_decode

    def add_request(self, prompt, sampling_params: SamplingParams | None = None) -> Sequence:
        if isinstance(prompt, str):
            if self.tokenizer is None:
                raise ValueError("string prompts need a tokenizer; pass token ids instead")
            prompt = self.tokenizer.encode(prompt)
        if not prompt:
            raise ValueError("empty prompt")
        seq = Sequence(prompt, sampling_params)
        # positions beyond the rotary table would be clamped by the kernels, and a request larger than the
        # whole cache could never be scheduled: refuse both here
        total = len(seq.prompt_tokens) + seq.sampling_params.max_tokens
        max_pos = getattr(self.runner.cfg, "max_position_embeddings", None)
        if max_pos is not None and total > max_pos:
            raise ValueError(f"prompt ({len(seq.prompt_tokens)}) + max_tokens ({seq.sampling_params.max_tokens}) "
                             f"exceeds max_position_embeddings ({max_pos})")
        bs = self.scheduler.block_size
        if -(-total // bs) > self.runner.num_kvcache_blocks:
            raise ValueError(f"request needs {-(-total // bs)} KV-cache blocks, the cache has "
                             f"{self.runner.num_kvcache_blocks}")
        self.scheduler.add(seq)
        return seq

    def is_finished(self) -> bool:
        return self.scheduler.is_finished()

    def step(self):
        """One schedule -> forward -> update round.
        Returns (scheduled, sampled tokens, finished, whether the step had decode rows only)."""
        seqs, _ = self.scheduler.schedule()
        if not seqs:
            raise ValueError("no sequences are scheduled")
        decode_only = all(_is_decode_row(s) for s in seqs)
        tokens = self.runner.run(seqs)
        finished = self.scheduler.update_from_output(seqs, tokens)
        return seqs, tokens, finished, decode_only

    def generate(self, prompts, sampling_params=None):
        """prompts: list of token-id lists (or strings with a tokenizer).  Returns (outputs, metrics);
        outputs[i] = {"token_ids": [...], "text": str | None, "num_preemptions": times the request's blocks were
        taken away and its K/V recomputed, "cached_tokens": prompt tokens taken from the
        prefix cache} in prompt order."""
        if not isinstance(sampling_params, (list, tuple)):
            sampling_params = [sampling_params] * len(prompts)
        seqs = [self.add_request(p, sp) for p, sp in zip(prompts, sampling_params)]
        metrics = GenerationMetrics(num_prompt_tokens=sum(len(s.prompt_tokens) for s in seqs))
        cuda = self.runner.device.type == "cuda"
        t_start = time.perf_counter()
        inflight = None                      # pipelined decode: (sequences, handle) of the step on the device
        while not self.is_finished() or inflight is not None:
            t0 = time.perf_counter()
            batch = self.scheduler.schedule()[0] if not self.is_finished() else []
            if self.pipelined_decode and batch and all(_is_decode_row(s) for s in batch):
                handle = self.runner.launch_decode(batch, inflight[1] if inflight else None)
                if inflight is not None:     # read step t while step t+1 runs
                    self.scheduler.resolve_pending(inflight[0], self.runner.fetch_tokens(inflight[1]))
                self.scheduler.advance_pending(batch)
                inflight = (batch, handle)
                metrics.step_times.append((time.perf_counter() - t0, len(batch), True))
                continue
            waited = inflight
            if inflight is not None:         # a step with prompt rows needs the token values on the host
                self.scheduler.resolve_pending(inflight[0], self.runner.fetch_tokens(inflight[1]))
                inflight = None
                kept = [s for s in batch if not s.is_finished]      # minus sequences that just hit EOS
                for s in batch:
                    if s.is_finished:
                        s.num_scheduled_tokens = 0
                batch = kept
            if not batch:
                if waited is None and not self.is_finished():
                    # nothing in flight, nothing schedulable, requests left: cannot make progress
                    raise RuntimeError("scheduler made no progress: the KV cache cannot hold the next chunk of any "
                                       "request (num_kvcache_blocks too small for the admitted requests)")
                continue
            decode_only = all(_is_decode_row(s) for s in batch)
            tokens = self.runner.run(batch)
            self.scheduler.update_from_output(batch, tokens)
            if cuda:
                torch.cuda.synchronize(self.runner.device)
            now = time.perf_counter()
            metrics.step_times.append((now - t0, sum(t is not None for t in tokens), decode_only))
            for s, t in zip(batch, tokens):
                if t is not None and len(s.output_tokens) == 1:
                    metrics.first_token_time[s.id] = now - t_start
        metrics.num_generated_tokens = sum(len(s.output_tokens) for s in seqs)
        metrics.wall_time = time.perf_counter() - t_start
        outs = []
        for s in seqs:
            text = self.tokenizer.decode(s.output_tokens) if self.tokenizer is not None else None
            outs.append({"token_ids": list(s.output_tokens), "text": text, "cached_tokens": s.num_cached_tokens,
                         "num_preemptions": s.num_preemptions})
        return outs, metrics
