#!/bin/bash
# configs[2]'s one-GPU share (B = 1024, P = 512): tile / schedule A/B, interleaved on one box
O=gpurun_out/r04_c3; mkdir -p $O
python tools/ab_tune.py --rounds 3 --out $O/c3_ab.txt \
  -v "base:" -v "fused:--tune backward=fused" -v "fork:--tune backward=fork" \
  -v "tile21:--tune proj_tile=21" -v "tile21_fused:--tune proj_tile=21 --tune backward=fused" \
  -v "tile21_lds57:--tune proj_tile=21 --tune k1_share_lds_kb=57" -v "tnt:--tune tn_transpose=1" \
  -v "nolds:--tune k1_share_lds_kb=-1" \
  -- --config c3 --steps 200 --warmup 20 2>&1 | tail -10
python tools/ab_tune.py --rounds 2 --out $O/c3_sim_ab.txt \
  -v "sim0:--simulate-allreduce-us 0" -v "sim60:--simulate-allreduce-us 60" -v "sim60_split:--simulate-allreduce-us 60 --dp-split" \
  -v "sim60_t21:--simulate-allreduce-us 60 --tune proj_tile=21" \
  -- --config c3 --steps 200 --warmup 20 2>&1 | tail -6
