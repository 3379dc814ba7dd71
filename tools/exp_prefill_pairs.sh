#!/bin/bash
# prefill launches whose (sequence, kv head) pairs do not fill the 8 XCDs evenly: the built library against a
# variant library (nano-vllm_amd/lib/variants/lib_*.so), same process per case
CASES=tp1_chunk,tp2_chunk,tp4_chunk,tp8_chunk,tp1_full,tp2_full,tp4_full,tp8_full,pairs12_full,pairs7_full,pairs24_full,cfg3_7seq_chunks,b4_n8192_h32
echo "== built library"; timeout -k 10 300 python3 tools/prefill_bench.py --kernels auto --rounds 3 --iters 4 --cases $CASES 2>/dev/null
for lib in nano-vllm_amd/lib/variants/lib_*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"; timeout -k 10 300 python3 tools/prefill_bench.py --lib $lib --kernels auto --rounds 3 --iters 4 --cases $CASES 2>/dev/null
done
