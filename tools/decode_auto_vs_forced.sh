#!/bin/bash
# decode attention launch at the small per-GPU shapes: what the launch plan picks ("auto") next to the forced forms
set -e
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5"
echo "== H 8/1 shard, B 256, ctx 2048"; $S --hq 8 --hkv 1 --batch 256 --ctx-lo 2048 --ctx-hi 2048 --layers 8 --modes auto,wgs0,wgs1 --chunks 0,-1024 2>/dev/null
echo "== H 16/2 shard, B 256, ctx 2048"; $S --hq 16 --hkv 2 --batch 256 --ctx-lo 2048 --ctx-hi 2048 --layers 8 --modes auto,wgs0,wgs1 --chunks 0,-1024 2>/dev/null
echo "== 4 x 32k, H 32/8"; $S --hq 32 --hkv 8 --batch 4 --ctx-lo 32768 --ctx-hi 32768 --layers 4 --modes auto,wgs0,wgs1 --chunks 0,-1024 2>/dev/null
