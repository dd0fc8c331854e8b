#!/bin/bash
# K-split 96-row projection tiles (tt_tuning.proj_tile = 33 / 32) against the default 64x64 tiles
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s16; mkdir -p $out
cd $root
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "ksplit or metric_shape or projection_fwd_bwd" > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc" >> $out/tests.log; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
python tools/ab_tune.py --rounds 2 --out $out/c2_cache.txt -v "base:" -v "ks33:--tune proj_tile=33" -- --config c2 --inputs cache --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/c2_default.txt -v "base:" -v "ks33:--tune proj_tile=33" -- --config c2 --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/c2_x60.txt -v "base:" -v "ks33:--tune proj_tile=33" -- --config c2 --simulate-allreduce-us 60 --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/c3_cache.txt -v "base:" -v "ks32:--tune proj_tile=32" -- --config c3 --inputs cache --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/c3_single.txt -v "base:" -v "ks32:--tune proj_tile=32" -- --config c3 --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/c3_x60.txt -v "base:" -v "ks32:--tune proj_tile=32" -- --config c3 --simulate-allreduce-us 60 --steps 200 --warmup 20 2>&1 | tail -3
