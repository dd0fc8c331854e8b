#!/bin/bash
# usage (on the GPU box): tools/pmc_k1.sh CONFIG ALG_BYTES KERNELS  — two rocprofv3 --pmc passes of `bench.py --k1-only`
cfg=$1; alg=$2; kern=$3
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/pmc_fetch_$cfg -o f -- python3 $root/bench.py --config $cfg --k1-only --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/pmc_write_$cfg -o w -- python3 $root/bench.py --config $cfg --k1-only --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2>&1 || exit 1
python3 $root/tools/pmc_traffic.py $root/gpurun_out/pmc_fetch_$cfg $root/gpurun_out/pmc_write_$cfg $root/gpurun_out/k1_traffic_$cfg.json $alg $kern
