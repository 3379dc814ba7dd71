set -e
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 6 --modes auto --chunks 0"
for extra in "" "--no-cu-q"; do
echo "#### extra=$extra"
echo "== H 8/1"; $S --hq 8 --hkv 1 --batch 256 --ctx-lo 2048 --ctx-hi 2048 --layers 8 $extra 2>/dev/null
echo "== H 16/2"; $S --hq 16 --hkv 2 --batch 256 --ctx-lo 2048 --ctx-hi 2048 --layers 8 $extra 2>/dev/null
echo "== 4 x 32k"; $S --hq 32 --hkv 8 --batch 4 --ctx-lo 32768 --ctx-hi 32768 --layers 4 $extra 2>/dev/null
echo "== 1 x 32k"; $S --hq 32 --hkv 8 --batch 1 --ctx-lo 32768 --ctx-hi 32768 --layers 8 $extra 2>/dev/null
echo "== headline"; $S $extra 2>/dev/null
done
