#!/bin/bash
# decode launch on skewed batches: most sequences short, a few long; plan's choice (from the longest length alone)
# against forced chunk sizes and the chunk nv_attn_decode_suggest_chunk derives from the sum of the lengths
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5 --modes auto --layers 8 --hint --chunk-by-call"
echo "== 256 x U[513,767] + 1 x 8192"; $S --batch 257 --long 1:8192 --chunks 0,512,1024,2048 2>/dev/null
echo "== 256 x U[513,767] + 4 x 4096"; $S --batch 260 --long 4:4096 --chunks 0,512,1024,2048 2>/dev/null
echo "== 256 x U[513,767] + 1 x 32768"; $S --batch 257 --long 1:32768 --chunks 0,512,1024,2048 2>/dev/null
echo "== 256 x U[129,383] + 8 x 2048"; $S --batch 264 --ctx-lo 129 --ctx-hi 383 --long 8:2048 --chunks 0,256,512,1024 2>/dev/null
echo "== 256 x U[1024,3072]"; $S --layers 4 --ctx-lo 1024 --ctx-hi 3072 --chunks 0,1024 2>/dev/null
echo "== H 64/8: 256 x U[513,767] + 1 x 8192"; $S --layers 4 --hq 64 --hkv 8 --batch 257 --long 1:8192 --chunks 0,512,1024 2>/dev/null
