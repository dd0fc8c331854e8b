#!/bin/bash
# configs[3]: ONE gather workgroup per CU with many rows in flight (so that the projection tiles keep their residency)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/c4trace2; mkdir -p $out
cd $root
rm -f $out/trace.txt
for cfg in "TT_K1_STREAM_WGS=2 TT_K1_STREAM_U=6" "TT_K1_STREAM_WGS=1 TT_K1_STREAM_U=8" "TT_K1_STREAM_WGS=1 TT_K1_STREAM_U=10" "TT_K1_STREAM_WGS=1 TT_K1_STREAM_U=12" "TT_K1_STREAM_WGS=2 TT_K1_STREAM_U=6"; do
  echo "== $cfg" >> $out/trace.txt
  env $cfg python3 bench.py --config c4 --no-cpu-baseline --steps 30 --warmup 8 2>> $out/trace.txt | python3 -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["ms_per_step"],3), "K1 in-loop ms", round(r["avg_launch_ms"],3), "alone", round(r["alone"]["avg_launch_ms"],3), j["calibration"]["ms_per_step"])' >> $out/trace.txt
done
grep -v amdgpu.ids $out/trace.txt
