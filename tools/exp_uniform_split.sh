#!/bin/bash
# what splitting costs a batch of EQUAL long sequences (where it buys nothing): plan's choice against forced chunks
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5 --modes auto --variants default"
echo "== 256 x 2048, H 16/8"; $S --layers 4 --ctx-lo 2048 --ctx-hi 2048 --chunks 0,1024,512 2>/dev/null
echo "== 256 x 2048, H 64/8"; $S --layers 4 --hq 64 --hkv 8 --ctx-lo 2048 --ctx-hi 2048 --chunks 0,1024,512 2>/dev/null
echo "== 64 x 8192, H 16/8"; $S --layers 4 --batch 64 --ctx-lo 8192 --ctx-hi 8192 --chunks 0,1024,512 2>/dev/null
echo "== 256 x U[1024,3072], H 16/8"; $S --layers 4 --ctx-lo 1024 --ctx-hi 3072 --chunks 0,2048,1024,512 2>/dev/null
echo "== 256 x U[257,1023], H 16/8"; $S --layers 8 --ctx-lo 257 --ctx-hi 1023 --chunks 0,512,256 2>/dev/null
