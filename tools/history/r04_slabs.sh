#!/bin/bash
O=gpurun_out/r04_slabs; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_model_api.py -x -q -m gpu -k "engine or trainer or metric_size" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python tools/ab_tune.py --rounds 3 --out $O/c2_slabs.txt -v "s8:--grad-slabs 8" -v "s7:--grad-slabs 7" -v "s6:--grad-slabs 6" -v "s4:--grad-slabs 4" -v "s10:--grad-slabs 10" \
   -- --steps 200 --warmup 20 2>&1 | tail -7
python tools/ab_tune.py --rounds 2 --out $O/misc_slabs.txt -v "s2:--grad-slabs 2" -v "s3:--grad-slabs 3" -v "s4:--grad-slabs 4" -- --batch 512 2>&1 | tail -4
python tools/ab_tune.py --rounds 2 --out $O/misc_slabs.txt -v "s16:--grad-slabs 16" -v "s7:--grad-slabs 7" -v "s8:--grad-slabs 8" -- --batch 4096 2>&1 | tail -4
python tools/ab_tune.py --rounds 2 --out $O/misc_slabs.txt -v "s4:--grad-slabs 4" -v "s3:--grad-slabs 3" -v "s5:--grad-slabs 5" -- --config c3 2>&1 | tail -4
python tools/ab_tune.py --rounds 2 --out $O/misc_slabs.txt -v "s8:--grad-slabs 8" -v "s7:--grad-slabs 7" -- --inputs cache 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $O/misc_slabs.txt -v "s16:--grad-slabs 16" -v "s7:--grad-slabs 7" -- --loss inbatch --batch 4096 --steps 100 2>&1 | tail -3
