#!/bin/bash
# uneven decode batches: the host's length hint (row "chunk= N" against "chunk= 0") and longest-first row order
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5 --modes auto --chunks 0 --hint --chunk-by-call"
for B in 128 256 512 1024; do for srt in "" desc; do
  L=2; [ $B -le 256 ] && L=4
  echo "B $B U[100,2000] order '${srt:-random}':"; $S --layers $L --batch $B --ctx-lo 100 --ctx-hi 2000 ${srt:+--sort $srt} 2>/dev/null | grep "chunk" | cut -c1-90
done; done
