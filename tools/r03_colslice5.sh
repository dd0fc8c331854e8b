#!/bin/bash
# Round 3: column-sliced K1 built with at most 64 registers (8 waves per SIMD; libtt_hotpath_ab.so, -DTT_COL_WAVES_PER_EU=8)
# against the default build (66 registers, 7 waves per SIMD) beside the projection chain.
# (historical record: the build-time switch this A/B needed was removed from gather_pool.hip after the measurement.)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/colslice; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"] or {}; print(round(j["ms_per_step"]*1e3,1), "us  K1 in-loop", round((r.get("avg_launch_ms") or 0)*1e3,1), " alone", round(((r.get("alone") or {}).get("avg_launch_ms") or 0)*1e3,1))'
AB=$root/two-towers-overlords_amd/libtt_hotpath_ab.so
rm -f $out/ab5.txt
for rep in 1 2; do
echo "66 registers, 8 waves, default reservation: $(python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab5.txt
for w in 4 6 8 12 16; do for kb in 0 26; do
echo "64 registers, $w waves, reservation $kb KB: $(TT_HOTPATH_LIB=$AB TT_K1_COL_WAVES=$w TT_K1_SHARE_LDS_KB=$kb python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab5.txt
done; done
echo "64 registers, 8 waves, default reservation: $(TT_HOTPATH_LIB=$AB python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab5.txt
echo "64 registers, 8 waves, one stream: $(TT_HOTPATH_LIB=$AB python3 bench.py --no-cpu-baseline --no-secondary --no-overlap | python3 -c "$pick")" >> $out/ab5.txt
done
sort $out/ab5.txt
