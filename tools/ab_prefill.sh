#!/bin/bash
# A/B timing on one box: the built library against every library under nano-vllm_amd/lib/variants/, twice each.
CASES=${1:-chunk_512x2048,chunk_512x4096,chunk_512x8192,cfg3_7seq_chunks,b4_n8192_h32,b16_n2048_h32}
for rep in 1 2; do
  echo "== new (rep $rep)"; timeout -k 5 200 python -u tools/prefill_bench.py --kernels auto --rounds 3 --iters 4 --cases $CASES 2>/dev/null || exit 1
  for f in nano-vllm_amd/lib/variants/lib_*.so; do
    echo "== $f (rep $rep)"; timeout -k 5 200 python -u tools/prefill_bench.py --kernels auto --rounds 3 --iters 4 --cases $CASES --lib $f 2>/dev/null || exit 1
  done
done
