#!/bin/bash
# 64 x 32 projection / weight-gradient tiles (two waves per workgroup): parity, then interleaved A/B
O=gpurun_out/r04_n32; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "projection or engine_matches or awkward" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python tools/ab_tune.py --rounds 3 --out $O/c3_n32.txt -v "base:" -v "p10:--tune proj_tile=10" -v "p10_t10:--tune proj_tile=10 --tune tn_tile=10" -v "t10:--tune tn_tile=10" \
   -- --config c3 --steps 200 --warmup 20 2>&1 | tail -6
python tools/ab_tune.py --rounds 3 --out $O/c2_n32.txt -v "base:" -v "p10:--tune proj_tile=10" -v "p10_t10:--tune proj_tile=10 --tune tn_tile=10" -v "t10:--tune tn_tile=10" \
   -- --steps 200 --warmup 20 2>&1 | tail -6
python tools/ab_tune.py --rounds 2 --out $O/misc_n32.txt -v "base:" -v "p10:--tune proj_tile=10" -v "p10_t10:--tune proj_tile=10 --tune tn_tile=10" -- --batch 512 2>&1 | tail -4
python tools/ab_tune.py --rounds 2 --out $O/misc_n32.txt -v "base:" -v "p10:--tune proj_tile=10" -v "p10_t10:--tune proj_tile=10 --tune tn_tile=10" -- --inputs cache 2>&1 | tail -4
python tools/ab_tune.py --rounds 2 --out $O/misc_n32.txt -v "base:" -v "p10:--tune proj_tile=10" -- --config c3 --simulate-allreduce-us 60 2>&1 | tail -3
