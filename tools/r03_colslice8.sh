#!/bin/bash
# Round 3: column-sliced K1 with scalar-base addressing (62 registers: 8 waves per SIMD, 4 workgroups of 8 waves per CU)
# — parity, then the LDS reservation beside the chain again (4 x 39 KB would take the whole LDS).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/colslice; mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or metric_size" > $out/pytest8.txt 2>&1; rc=$?
tail -3 $out/pytest8.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"] or {}; print(round(j["ms_per_step"]*1e3,1), "us  K1 in-loop", round((r.get("avg_launch_ms") or 0)*1e3,1), " alone", round(((r.get("alone") or {}).get("avg_launch_ms") or 0)*1e3,1))'
rm -f $out/ab8.txt
for rep in 1 2; do
echo "default reservation: $(python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab8.txt
for kb in 0 20 26 32 37 44 53; do
echo "reservation $kb KB: $(TT_K1_SHARE_LDS_KB=$kb python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab8.txt
done
echo "4 waves, default reservation: $(TT_K1_COL_WAVES=4 python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab8.txt
echo "one stream: $(python3 bench.py --no-cpu-baseline --no-secondary --no-overlap | python3 -c "$pick")" >> $out/ab8.txt
echo "one stream 4 waves: $(TT_K1_COL_WAVES=4 python3 bench.py --no-cpu-baseline --no-secondary --no-overlap | python3 -c "$pick")" >> $out/ab8.txt
done
sort $out/ab8.txt
