#!/bin/bash
# configs[1]: wave priority (s_setprio) of the projection kernels x backward schedule — interleaved repetitions
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/prioc2; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; c=j["calibration"]; print(round(j["ms_per_step"]*1e3,1), "us  K1 in-loop", round(r["avg_launch_ms"]*1e3,1), "calibrate picked", c["backward"], {k: round(v*1e3,1) for k,v in c["ms_per_step"].items()})'
rm -f $out/ab3.txt
for rep in 1 2 3; do
for p in 0 3; do for bwd in fused fork auto; do
  echo "TT_GEMM_PRIO=$p TT_BWD=$bwd: $(TT_GEMM_PRIO=$p TT_BWD=$bwd python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab3.txt
done; done; done
sort $out/ab3.txt
