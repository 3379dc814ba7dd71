#!/usr/bin/env python3
"""Tuning aid: wall time vs GPU time (events) of the chunked-prefill engine steps of config 2."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.e2e_bench import _build  # noqa: E402
from nanovllm_amd.engine.llm_engine import LLMEngine  # noqa: E402
from nanovllm_amd.engine.sequence import SamplingParams  # noqa: E402


def main(batch=64, prompt_len=512):
    dev = "cuda:0"
    cfg, model = _build("qwen3_0_6b", dev, 0)
    eng = LLMEngine(model, cfg, block_size=256, num_kvcache_blocks=batch * 4 + 8, device=dev, eos=-1, use_graphs=False)
    g = torch.Generator().manual_seed(0)
    for rnd in range(2):                              # round 0 warms up
        prompts = [torch.randint(0, cfg.vocab_size, (prompt_len,), generator=g).tolist() for _ in range(batch)]
        for p in prompts:
            eng.add_request(p, SamplingParams(0.0, 1))
        walls, gpus = [], []
        while not eng.is_finished():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            a.record()
            eng.step()
            b.record()
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
            gpus.append(a.elapsed_time(b))
        if rnd:
            print(f"{len(walls)} prefill steps: wall {1e3 * sum(walls) / len(walls):.3f} ms/step, "
                  f"event span {sum(gpus) / len(gpus):.3f} ms/step")
            if os.environ.get("NV_STEP_PROFILE") == "1":
                import cProfile
                import pstats
                prompts = [torch.randint(0, cfg.vocab_size, (prompt_len,), generator=g).tolist() for _ in range(batch)]
                for p in prompts:
                    eng.add_request(p, SamplingParams(0.0, 1))
                pr = cProfile.Profile()
                pr.enable()
                while not eng.is_finished():
                    eng.step()
                torch.cuda.synchronize()
                pr.disable()
                pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
