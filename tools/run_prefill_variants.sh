#!/bin/bash
# bench every library under nano-vllm_amd/lib/variants/lib_pf_*.so plus the default one on one case
CASE=${1:-b4_n8192_h32}
echo "default"; timeout -k 5 100 python -u tools/prefill_bench.py --kernels 1 --rounds 3 --iters 3 --cases $CASE || exit 1
for f in nano-vllm_amd/lib/variants/lib_pf_*.so; do
  echo "$f"; timeout -k 5 100 python -u tools/prefill_bench.py --kernels 1 --rounds 3 --iters 3 --cases $CASE --lib $f || exit 1
done
