#!/bin/bash
# short chunks of one sequence (512 queries over 512 .. 2048 keys): both kernels, forced key-range split counts
for sp in 0 2 3 4; do echo "split $sp"; timeout -k 5 200 python -u tools/prefill_bench.py --kernels 0,1 --rounds 3 --iters 4 --split $sp --cases chunk_512x512,chunk_512x1024,chunk_512x1536,chunk_512x2048 2>/dev/null || exit 1; done
echo "auto"; timeout -k 5 200 python -u tools/prefill_bench.py --kernels auto --rounds 3 --iters 4 --cases chunk_512x512,chunk_512x1024,chunk_512x1536,chunk_512x2048 2>/dev/null
