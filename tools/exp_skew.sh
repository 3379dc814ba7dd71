#!/bin/bash
# decode launch on skewed batches: most sequences short, a few long; plan's choice against forced chunk sizes
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5 --modes auto --layers 8"
echo "== 256 x U[513,767] + 1 x 8192"; $S --long 1:8192 --chunks 0,512,1024,2048 2>/dev/null
echo "== 256 x U[513,767] + 4 x 4096"; $S --long 4:4096 --chunks 0,512,1024,2048 2>/dev/null
echo "== 256 x U[513,767] + 1 x 32768"; $S --long 1:32768 --chunks 0,512,1024,2048 2>/dev/null
echo "== 256 x U[129,383] + 8 x 2048"; $S --ctx-lo 129 --ctx-hi 383 --long 8:2048 --chunks 0,256,512,1024 2>/dev/null
