for CASE in chunk_512x4096 chunk_512x8192; do
echo "== $CASE default"; timeout -k 5 100 python -u tools/prefill_bench.py --kernels 1 --rounds 5 --iters 8 --cases $CASE 2>/dev/null || exit 1
for f in nano-vllm_amd/lib/variants/lib_pf_*.so; do
  echo "$f"; timeout -k 5 100 python -u tools/prefill_bench.py --kernels 1 --rounds 5 --iters 8 --cases $CASE --lib $f 2>/dev/null || exit 1
done; done
