#!/bin/bash
# Round 3: column-sliced K1 (8 waves per workgroup) beside the projection chain — the gather workgroup's LDS
# reservation (TT_K1_SHARE_LDS_KB; default: the 64x64 tile's LDS + 3 KB = 39 KB) and the chain's tile.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/colslice; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"] or {}; print(round(j["ms_per_step"]*1e3,1), "us  K1 in-loop", round((r.get("avg_launch_ms") or 0)*1e3,1), " alone", round(((r.get("alone") or {}).get("avg_launch_ms") or 0)*1e3,1))'
rm -f $out/ab4.txt
for rep in 1 2; do
echo "default reservation: $(python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab4.txt
for kb in 0 16 26 32 37 44 53 64 80; do
echo "reservation $kb KB: $(TT_K1_SHARE_LDS_KB=$kb python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab4.txt
done
for kb in 39 53 64; do
echo "128x64 tiles, reservation $kb KB: $(TT_TILE=21 TT_K1_SHARE_LDS_KB=$kb python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab4.txt
done
done
sort $out/ab4.txt
