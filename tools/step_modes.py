#!/usr/bin/env python3
"""The bench's decode step (28 layers x 4 launches) timed four ways on one box: launched one by one with a
HIP event pair around every attention launch, with a pair around every 7th (what bench.py's timed region
does), without events, and replayed as one captured HIP graph (what the engine's decode step does).

    python tools/step_modes.py [--steps 20] [--rounds 5]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def timed(fn, steps, dev):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--two-launch", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        w = bench.Workload(dev, 28, 256, 513, 768, seed=0)
        plan = w.build_plan(False, not a.two_launch)   # launches go to `side` (the current stream at plan time)
        for _ in range(3):
            bench.run_step(plan)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            bench.run_step(plan)
        rows = {"events": [], "sampled": [], "plain": [], "graph": []}
        for _ in range(a.rounds):
            ev = []
            rows["events"].append(timed(lambda: bench.run_step(plan, ev, every=1), a.steps, dev))
            rows["sampled"].append(timed(lambda: bench.run_step(plan, ev, phase=len(ev)), a.steps, dev))
            rows["plain"].append(timed(lambda: bench.run_step(plan), a.steps, dev))
            rows["graph"].append(timed(g.replay, a.steps, dev))
    nbytes = w.step_bytes(False, not a.two_launch)
    for k, v in rows.items():
        v.sort()
        med = v[len(v) // 2]
        print(f"{k:7s} ms/step median {med:.4f}  (min {v[0]:.4f} max {v[-1]:.4f})  "
              f"{nbytes / med / 1e6:.1f} GB/s = {nbytes / med / 1e6 / 80:.2f} % of 8 TB/s")


if __name__ == "__main__":
    main()
