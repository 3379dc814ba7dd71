#!/bin/bash
# decode attention launch at the small per-GPU shapes of BASELINE configs 4 and 5: work-item form x wave target,
# for the built library and every variant under nano-vllm_amd/lib/variants/
set -e
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 4"
echo "== H 8/1 shard, B 256, ctx 2048"; $S --hq 8 --hkv 1 --batch 256 --ctx-lo 2048 --ctx-hi 2048 --layers 8 --modes wgs0,wgs1 --chunks 0,-1024,-2048 2>/dev/null
echo "== H 16/2 shard, B 256, ctx 2048"; $S --hq 16 --hkv 2 --batch 256 --ctx-lo 2048 --ctx-hi 2048 --layers 8 --modes wgs0,wgs1 --chunks 0,-1024,-2048 2>/dev/null
echo "== H 32/4 shard, B 256, ctx 2048"; $S --hq 32 --hkv 4 --batch 256 --ctx-lo 2048 --ctx-hi 2048 --layers 4 --modes auto --chunks 0 2>/dev/null
echo "== 4 x 32k, H 32/8"; $S --hq 32 --hkv 8 --batch 4 --ctx-lo 32768 --ctx-hi 32768 --layers 4 --modes wgs0,wgs1 --chunks 0,-1024,-2048 2>/dev/null
echo "== 8 x 32k, H 32/8"; $S --hq 32 --hkv 8 --batch 8 --ctx-lo 32768 --ctx-hi 32768 --layers 2 --modes wgs0,wgs1 --chunks 0 2>/dev/null
echo "== 1 x 32k, H 32/8"; $S --hq 32 --hkv 8 --batch 1 --ctx-lo 32768 --ctx-hi 32768 --layers 8 --modes wgs0,wgs1,static --chunks 0,-1024 2>/dev/null
echo "== headline"; $S --modes auto --chunks 0 2>/dev/null
