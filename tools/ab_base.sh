# A/B on one box: the built library against nano-vllm_amd/lib/variants/lib_base.so, twice each
CASES=${1:-chunk_512x2048,chunk_512x4096,chunk_512x8192,cfg3_7seq_chunks,b4_n8192_h32}
for rep in 1 2; do
  echo "== new (rep $rep)"; timeout -k 5 200 python -u tools/prefill_bench.py --kernels auto --rounds 5 --iters 8 --cases $CASES 2>/dev/null || exit 1
  echo "== base (rep $rep)"; timeout -k 5 200 python -u tools/prefill_bench.py --kernels auto --rounds 5 --iters 8 --cases $CASES --lib nano-vllm_amd/lib/variants/lib_base.so 2>/dev/null || exit 1
done
