#!/bin/bash
# rows longest first, one sequence 2 .. 6 x the mean: from which skew on does the suggested chunk pay?
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5 --modes auto --chunks 0,512,1024 --hint --chunk-by-call --sort desc --layers 2"
run() { echo "$*:"; $S "$@" 2>/dev/null | grep "chunk" | cut -c1-90; }
for B in 64 256 512; do for long in 3000 4000 6000; do
  run --batch $B --ctx-lo 100 --ctx-hi 2000 --long 1:$long
done; done
for B in 64 256; do for long in 1500 2000 3000; do
  run --batch $B --long 1:$long
done; done
