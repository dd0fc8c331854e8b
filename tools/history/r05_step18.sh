#!/bin/bash
# K-split tiles v2 at batch sizes where no panel straddles the towers (2016 = 21 x 96, 960 = 10 x 96): what the step would
# gain if a straddling panel cost nothing extra
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s18; mkdir -p $out
cd $root
python tools/ab_tune.py --rounds 2 --out $out/c2_cache.txt -v "base:" -v "ks33:--tune proj_tile=33" -- --config c2 --batch 2016 --inputs cache --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/c2_default.txt -v "base:" -v "ks33:--tune proj_tile=33" -v "ks33_lds:--tune proj_tile=33 --tune k1_share_lds_kb=51" -- --config c2 --batch 2016 --steps 200 --warmup 20 2>&1 | tail -4
python tools/ab_tune.py --rounds 2 --out $out/c2_x60.txt -v "base:" -v "ks33:--tune proj_tile=33" -- --config c2 --batch 2016 --simulate-allreduce-us 60 --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/c3_single.txt -v "base:" -v "ks32:--tune proj_tile=32" -- --config c3 --batch 960 --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/c3_x60.txt -v "base:" -v "ks32:--tune proj_tile=32" -- --config c3 --batch 960 --simulate-allreduce-us 60 --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/c3_x60_full.txt -v "base:" -v "ks32:--tune proj_tile=32" -- --config c3 --simulate-allreduce-us 60 --steps 200 --warmup 20 2>&1 | tail -3
