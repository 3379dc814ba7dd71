#!/usr/bin/env python3
"""Tuning aid: GPU time of one captured decode step (HIP-graph replay, events) next to the wall time
of a full engine step, for the config-2 model and batch."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.e2e_bench import _build  # noqa: E402
from nanovllm_amd.engine.llm_engine import LLMEngine  # noqa: E402
from nanovllm_amd.engine.sequence import SamplingParams  # noqa: E402


def main(model_name="qwen3_0_6b", batch=256, prompt_len=512):
    dev = "cuda:0"
    cfg, model = _build(model_name, dev, 0)
    eng = LLMEngine(model, cfg, block_size=256, num_kvcache_blocks=batch * 4 + 8, device=dev, eos=-1, use_graphs=True)
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(0, cfg.vocab_size, (prompt_len,), generator=g).tolist() for _ in range(batch)]
    for p in prompts:
        eng.add_request(p, SamplingParams(0.0, 64))
    while any(s.num_computed_tokens < len(s.prompt_tokens) for s in eng.scheduler.running) or eng.scheduler.waiting:
        eng.step()
    for _ in range(5):
        eng.step()                                  # decode steps: graph captured and warm
    torch.cuda.synchronize()
    walls = []
    for _ in range(20):
        t0 = time.perf_counter()
        eng.step()
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    graph = next(iter(eng.runner._graphs.values()))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        graph.graph.replay()
    b.record()
    torch.cuda.synchronize()
    print(f"engine step wall {1e3 * sum(walls) / len(walls):.3f} ms; graph replay (GPU) {a.elapsed_time(b) / 20:.3f} ms")
    if os.environ.get("NV_STEP_PROFILE") == "1":
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(30):
            eng.step()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
