#!/bin/bash
# engine end to end on 512 prompts of 100 .. 2000 tokens: decode rows in the scheduler's order against longest first
set -e
for m in qwen3_0_6b qwen3_8b; do
  timeout -k 10 280 python -u tools/e2e_bench.py --model $m --scenario uneven_batch --scheduler-order
  timeout -k 10 280 python -u tools/e2e_bench.py --model $m --scenario uneven_batch
done
