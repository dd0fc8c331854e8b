#!/bin/bash
# per-step GPU times + streaming-K1 census at configs[3]
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/c4trace; mkdir -p $out
cd $root
for cfg in "TT_K1_STREAM_U=6" "TT_K1_STREAM_U=5" "TT_K1_STREAM_U=7" "TT_K1_STREAM_U=4" "TT_K1_STREAM_U=6 TT_K1_STREAM_LDS_KB=40"; do
  echo "== $cfg" >> $out/trace.txt
  env $cfg TT_TRACE=1 python3 bench.py --config c4 --no-cpu-baseline --steps 30 --warmup 8 2>> $out/trace.txt | python3 -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["ms_per_step"],3), "K1 in-loop ms", round(r["avg_launch_ms"],3), "alone", round(r["alone"]["avg_launch_ms"],3), j["calibration"]["ms_per_step"])' >> $out/trace.txt
done
grep -v amdgpu.ids $out/trace.txt
