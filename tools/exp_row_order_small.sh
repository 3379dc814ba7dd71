#!/bin/bash
# smaller uneven decode batches: row order (random / longest first) and the host's length hint
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 7 --modes auto --chunks 0 --hint --chunk-by-call --layers 4"
for hq in 16 32; do for B in 8 16 32 64 128; do for srt in "" desc; do
  echo "B $B heads $hq/8 U[100,2000] order '${srt:-random}':"; $S --hq $hq --batch $B --ctx-lo 100 --ctx-hi 2000 ${srt:+--sort $srt} 2>/dev/null | grep "chunk" | cut -c1-90
done; done; done
