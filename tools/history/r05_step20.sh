#!/bin/bash
# K-split tiles v4: LDS writes of the next K-step in the MFMA shadow, first products right after the barrier
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s20; mkdir -p $out
cd $root
timeout -k 10 120 ./build_tools/gemm_stamps > $out/stamps.txt 2>&1 < /dev/null; grep -v probe $out/stamps.txt | cut -c1-330
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "ksplit or metric_shape" > $out/tests.log 2>&1 < /dev/null; rc=$?; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
python tools/ab_tune.py --rounds 2 --out $out/c3_x60.txt -v "base:" -v "ks32:--tune proj_tile=32" -v "ks32_fork:--tune proj_tile=32 --tune backward=fork" -- --config c3 --simulate-allreduce-us 60 --steps 200 --warmup 20 2>&1 < /dev/null | tail -4
python tools/ab_tune.py --rounds 2 --out $out/c2_x60.txt -v "base:" -v "ks33:--tune proj_tile=33" -- --config c2 --simulate-allreduce-us 60 --steps 200 --warmup 20 2>&1 < /dev/null | tail -3
python tools/ab_tune.py --rounds 2 --out $out/c3_x60_960.txt -v "base:" -v "ks32:--tune proj_tile=32" -- --config c3 --batch 960 --simulate-allreduce-us 60 --steps 200 --warmup 20 2>&1 < /dev/null | tail -3
