#!/bin/bash
# the gather's slot fence taken from the engine's loss / fork event + the ready-wait ahead of Adam: parity, then A/B
O=gpurun_out/r04_fence; mkdir -p $O
python -m pytest tests/test_gpu_model_api.py tests/test_gpu_parity.py tests/test_gpu_dp.py -x -q -m gpu -k "trainer or ordered or copy_stream or token_store or run_training or two_ranks or rccl or inbatch" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python tools/ab_tune.py --rounds 3 --out $O/c2_fence.txt -v "old:--no-fence-at-loss" -v "new:" -v "new_fork:--tune backward=fork" -v "new_fused:--tune backward=fused" \
   -- --steps 200 --warmup 20 2>&1 | tail -6
python tools/ab_tune.py --rounds 2 --out $O/c3_fence.txt -v "old:--no-fence-at-loss" -v "new:" -v "new_fork:--tune backward=fork" \
   -v "old_sim60:--no-fence-at-loss --simulate-allreduce-us 60" -v "new_sim60:--simulate-allreduce-us 60" \
   -- --config c3 --steps 200 --warmup 20 2>&1 | tail -7
python tools/ab_tune.py --rounds 2 --out $O/misc_fence.txt -v "old:--no-fence-at-loss" -v "new:" -- --loss inbatch --batch 4096 --steps 100 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $O/misc_fence.txt -v "old:--no-fence-at-loss" -v "new:" -- --config c4 --steps 30 --warmup 8 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $O/misc_fence.txt -v "old:--no-fence-at-loss" -v "new:" -- --inputs index 2>&1 | tail -3
