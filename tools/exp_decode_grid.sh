#!/bin/bash
# decode launch over a grid of head shards x batch sizes (uniform 4096-key contexts): a scan for launch shapes
# that run far below their neighbours
for heads in "8 1" "16 2" "32 4" "64 8" "16 8"; do
  set -- $heads
  for B in 1 3 16 64; do
    echo -n "H $1/$2 B $B: "
    timeout -k 5 120 python -u tools/decode_sweep.py --rounds 4 --modes auto --chunks 0 --layers 4 --hq $1 --hkv $2 --batch $B --ctx-lo 4096 --ctx-hi 4096 2>/dev/null | grep "chunk=" | sed 's/ *default:auto chunk= *0 *//'
  done
done
