#!/bin/bash
# decode launch on SMALL uneven batches (fewer (sequence, kv head) pairs than the launch wants waves)
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5 --modes auto --layers 8 --hint --chunk-by-call"
echo "== H 32/8: 7 x 1024 + 1 x 32768"; $S --hq 32 --hkv 8 --batch 8 --ctx-lo 1024 --ctx-hi 1024 --long 1:32768 --chunks 0,256,512 2>/dev/null
echo "== H 16/8: 31 x U[300,700] + 1 x 16384"; $S --batch 32 --ctx-lo 300 --ctx-hi 700 --long 1:16384 --chunks 0,256,512 2>/dev/null
echo "== H 32/8: 8 x 32768 (uniform)"; $S --layers 2 --hq 32 --hkv 8 --batch 8 --ctx-lo 32768 --ctx-hi 32768 --chunks 0 2>/dev/null
echo "== H 64/8: 15 x U[1000,3000] + 1 x 20000"; $S --hq 64 --hkv 8 --batch 16 --ctx-lo 1000 --ctx-hi 3000 --long 1:20000 --chunks 0,512,1024 2>/dev/null
echo "== H 8/1: 63 x U[500,1500] + 1 x 12000"; $S --hq 8 --hkv 1 --batch 64 --ctx-lo 500 --ctx-hi 1500 --long 1:12000 --chunks 0,512,1024 2>/dev/null
