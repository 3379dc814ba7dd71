S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5 --modes auto --layers 8 --hint --chunk-by-call"
for B in 256 128 64; do
echo "== B=$B: U[513,767] incl. 1 x 8192"; $S --batch $B --long 1:8192 --chunks 0,512 2>/dev/null | grep chunk=
echo "== B=$B: U[513,767] incl. 1 x 32768"; $S --batch $B --long 1:32768 --chunks 0,1024 2>/dev/null | grep chunk=
done
echo "== B=256 H 64/8: incl. 2 x 8192"; $S --layers 4 --hq 64 --hkv 8 --batch 256 --long 2:8192 --chunks 0,1024 2>/dev/null | grep chunk=
