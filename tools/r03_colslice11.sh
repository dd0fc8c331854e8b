#!/bin/bash
# Round 3: column-sliced K1 beside the chain — 4-wave workgroups with a small LDS reservation (6-7 workgroups = 24-28 waves
# per CU in finer pieces) against the default 8-wave workgroups (3 per CU).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/colslice; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"] or {}; print(round(j["ms_per_step"]*1e3,1), "us  K1 in-loop", round((r.get("avg_launch_ms") or 0)*1e3,1), " alone", round(((r.get("alone") or {}).get("avg_launch_ms") or 0)*1e3,1))'
rm -f $out/ab11.txt
for rep in 1 2; do
echo "8 waves, default reservation: $(python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab11.txt
for kb in 0 16 22 26 30; do
echo "4 waves, reservation $kb KB: $(TT_K1_COL_WAVES=4 TT_K1_SHARE_LDS_KB=$kb python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab11.txt
done; done
sort $out/ab11.txt
