#!/bin/bash
# Round 5, step 11: balanced launches (tt_tuning.proj_balance): parity, then A/B in the chain-alone regimes and at the headline.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s11; mkdir -p $out
cd $root
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "projection or engine_step or golden or awkward" > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log; tail -3 $out/tests.log
T="--tune proj_balance=1"
python tools/ab_tune.py --rounds 2 --out $out/bal_c3_x60.txt -v "base:--tune backward=fork" -v "bal_fork:--tune backward=fork $T" -v "fused:--tune backward=fused" -v "bal_fused:--tune backward=fused $T" -- --config c3 --simulate-allreduce-us 60 --steps 200 --warmup 20 2>&1 | tail -5
python tools/ab_tune.py --rounds 2 --out $out/bal_c2_cache.txt -v "base:" -v "bal:$T" -v "bal_fork:$T --tune backward=fork" -- --config c2 --inputs cache --steps 200 --warmup 20 2>&1 | tail -4
python tools/ab_tune.py --rounds 2 --out $out/bal_c2.txt -v "base:" -v "bal:$T" -- --config c2 --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/bal_c3.txt -v "base:" -v "bal:$T" -- --config c3 --steps 200 --warmup 20 2>&1 | tail -3
echo done
