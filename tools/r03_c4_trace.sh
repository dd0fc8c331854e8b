#!/bin/bash
# configs[3]: wave priority of the streaming gather (s_setprio) beside the projection tiles
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/c4prio; mkdir -p $out
cd $root
rm -f $out/trace.txt
for cfg in "TT_K1_STREAM_PRIO=0" "TT_K1_STREAM_PRIO=3" "TT_K1_STREAM_PRIO=1" "TT_K1_STREAM_PRIO=3 TT_K1_STREAM_WGS=1 TT_K1_STREAM_U=8" "TT_K1_STREAM_PRIO=0" "TT_K1_STREAM_PRIO=3"; do
  echo "== $cfg" >> $out/trace.txt
  env $cfg python3 bench.py --config c4 --no-cpu-baseline --steps 30 --warmup 8 2>> $out/trace.txt | python3 -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["ms_per_step"],3), "K1 in-loop ms", round(r["avg_launch_ms"],3), "alone", round(r["alone"]["avg_launch_ms"],3), j["calibration"]["ms_per_step"])' >> $out/trace.txt
done
grep -v amdgpu.ids $out/trace.txt
