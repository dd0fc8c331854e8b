#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/last; mkdir -p $out
cd $root
bash tools/pmc_mfma_util.sh > $out/mfma.log 2>&1
tail -30 $out/mfma.log | grep -v amdgpu
( time python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_like.json ) 2> $out/bench_driver_like.time
python3 -c 'import sys,json; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=j["secondary"]; print(j["value"], j["ms_per_step"], "sec:", s["value"], s["ms_per_step"], s["roofline"]["frac"], s["roofline"]["alone"]["frac"], s["effective_warmup_steps"])' $out/bench_driver_like.json
tail -3 $out/bench_driver_like.time
