#!/bin/bash
# Round 3: column-sliced K1 beside the projection chain — waves per gather workgroup (the LDS reservation caps
# WORKGROUPS per CU, so this sets the gather's waves per CU).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/colslice; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"] or {}; print(round(j["ms_per_step"]*1e3,1), "us  K1 in-loop", round((r.get("avg_launch_ms") or 0)*1e3,1), " alone", round(((r.get("alone") or {}).get("avg_launch_ms") or 0)*1e3,1))'
rm -f $out/ab3.txt
for rep in 1 2; do
echo "c2 default row slices: $(TT_K1_COLS=0 python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab3.txt
for w in 4 6 8 12 16; do
echo "c2 default column slices, $w waves: $(TT_K1_COLS=1 TT_K1_COL_WAVES=$w python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab3.txt
done
for w in 8 12; do
echo "c2 index column slices, $w waves: $(TT_K1_COLS=1 TT_K1_COL_WAVES=$w python3 bench.py --no-cpu-baseline --no-secondary --inputs index | python3 -c "$pick")" >> $out/ab3.txt
echo "c3 share column slices, $w waves: $(TT_K1_COLS=1 TT_K1_COL_WAVES=$w python3 bench.py --no-cpu-baseline --no-secondary --config c3 | python3 -c "$pick")" >> $out/ab3.txt
echo "inbatch b4096 column slices, $w waves: $(TT_K1_COLS=1 TT_K1_COL_WAVES=$w python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096 | python3 -c "$pick")" >> $out/ab3.txt
echo "c2 ragged column slices, $w waves: $(TT_K1_COLS=1 TT_K1_COL_WAVES=$w python3 bench.py --no-cpu-baseline --no-secondary --ragged | python3 -c "$pick")" >> $out/ab3.txt
done
echo "c2 index row slices: $(TT_K1_COLS=0 python3 bench.py --no-cpu-baseline --no-secondary --inputs index | python3 -c "$pick")" >> $out/ab3.txt
echo "c3 share row slices: $(TT_K1_COLS=0 python3 bench.py --no-cpu-baseline --no-secondary --config c3 | python3 -c "$pick")" >> $out/ab3.txt
echo "inbatch b4096 row slices: $(TT_K1_COLS=0 python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096 | python3 -c "$pick")" >> $out/ab3.txt
echo "c2 ragged row slices: $(TT_K1_COLS=0 python3 bench.py --no-cpu-baseline --no-secondary --ragged | python3 -c "$pick")" >> $out/ab3.txt
done
sort $out/ab3.txt
