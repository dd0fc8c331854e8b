#!/bin/bash
# Round 3: column-sliced K1, same-box A/B of the two addressings: 62 registers (scalar base + 32-bit lane offset: 8 waves
# per SIMD) vs 66 registers (64-bit lane addresses: 7 waves per SIMD; libtt_hotpath_ab.so = -DTT_COL_ADDR64).
# (historical record: the build-time switch this A/B needed was removed from gather_pool.hip after the measurement.)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/colslice; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"] or {}; print(round(j["ms_per_step"]*1e3,1), "us  K1 in-loop", round((r.get("avg_launch_ms") or 0)*1e3,1), " alone", round(((r.get("alone") or {}).get("avg_launch_ms") or 0)*1e3,1))'
AB=$root/two-towers-overlords_amd/libtt_hotpath_ab.so
rm -f $out/ab9.txt
for rep in 1 2 3; do
echo "66 registers, default reservation: $(TT_HOTPATH_LIB=$AB python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab9.txt
echo "62 registers, default reservation: $(python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab9.txt
echo "62 registers, reservation 44 KB: $(TT_K1_SHARE_LDS_KB=44 python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab9.txt
echo "62 registers, reservation 48 KB: $(TT_K1_SHARE_LDS_KB=48 python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab9.txt
echo "66 registers, index inputs: $(TT_HOTPATH_LIB=$AB python3 bench.py --no-cpu-baseline --no-secondary --inputs index | python3 -c "$pick")" >> $out/ab9.txt
echo "62 registers, index inputs: $(python3 bench.py --no-cpu-baseline --no-secondary --inputs index | python3 -c "$pick")" >> $out/ab9.txt
echo "62 registers, index inputs, reservation 44 KB: $(TT_K1_SHARE_LDS_KB=44 python3 bench.py --no-cpu-baseline --no-secondary --inputs index | python3 -c "$pick")" >> $out/ab9.txt
done
sort $out/ab9.txt
