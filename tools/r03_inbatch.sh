#!/bin/bash
# Round 3: in-batch-negatives loss (BASELINE.json configs[4], B = 4096 per GPU) — parity, then tile / residency A/B.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/inbatch; mkdir -p $out
cd $root
python3 -m pytest tests -q -m gpu -k "inbatch or stream_policy" > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
tail -2 $out/pytest.txt
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"]*1e3,1), "us")'
B="python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096"
rm -f $out/ab.txt
for rep in 1 2; do
echo "default (128x128 S tiles): $(TT_INBATCH_TILE=2 $B | python3 -c "$pick")" >> $out/ab.txt
echo "TT_INBATCH_TILE=1 (64x64 S tiles): $(TT_INBATCH_TILE=1 $B | python3 -c "$pick")" >> $out/ab.txt
echo "no-overlap: $($B --no-overlap | python3 -c "$pick")" >> $out/ab.txt
echo "k1 cap 2: $($B --k1-blocks-per-cu 2 | python3 -c "$pick")" >> $out/ab.txt
echo "k1 cap 1: $($B --k1-blocks-per-cu 1 | python3 -c "$pick")" >> $out/ab.txt
echo "k1 cap 2, 64x64 S: $(TT_INBATCH_TILE=1 $B --k1-blocks-per-cu 2 | python3 -c "$pick")" >> $out/ab.txt
done
echo "b2048 default: $(python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch | python3 -c "$pick")" >> $out/ab.txt
cat $out/ab.txt
