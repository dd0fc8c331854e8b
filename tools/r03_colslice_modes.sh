#!/bin/bash
# Round 3: column-sliced K1 across input modes and batch sizes — Zipf ids (the row slices lose their balance there, the
# column slices cannot), small and large batches (the form-selection rule), against row slices and the per-text form.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/colslice; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"] or {}; print(round(j["ms_per_step"]*1e3,1), "us  K1 in-loop", round((r.get("avg_launch_ms") or 0)*1e3,1), " alone", round(((r.get("alone") or {}).get("avg_launch_ms") or 0)*1e3,1))'
rm -f $out/modes.txt
B="python3 bench.py --no-cpu-baseline --no-secondary"
for rep in 1 2; do
echo "zipf column slices: $($B --zipf | python3 -c "$pick")" >> $out/modes.txt
echo "zipf row slices: $(TT_K1_COLS=0 $B --zipf | python3 -c "$pick")" >> $out/modes.txt
echo "zipf per-text: $(TT_K1_SLICED=0 $B --zipf | python3 -c "$pick")" >> $out/modes.txt
for b in 256 512 1024 4096 8192; do
echo "B=$b column slices: $($B --batch $b | python3 -c "$pick")" >> $out/modes.txt
echo "B=$b row slices: $(TT_K1_COLS=0 $B --batch $b | python3 -c "$pick")" >> $out/modes.txt
echo "B=$b per-text: $(TT_K1_SLICED=0 $B --batch $b | python3 -c "$pick")" >> $out/modes.txt
done
done
sort $out/modes.txt
