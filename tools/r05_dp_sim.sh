#!/bin/bash
# The data-parallel step on ONE GPU with the RCCL all-reduce replaced by a stand-in kernel of X us (bench.py
# --simulate-allreduce-us): the real code path (slab reduce -> flat bucket -> exchange -> Adam from the bucket), at the
# per-GPU shapes BASELINE.json names for 8 GPUs.  NOT a scaling measurement: RCCL's own CU use is not modelled.
O=gpurun_out/r05_dp; mkdir -p $O
for spec in "c2:--config c2" "c3:--config c3" "inbatch512:--config c2 --loss inbatch --batch 512"; do
  tag=${spec%%:*}; args=${spec#*:}
  python tools/ab_tune.py --rounds 2 --out $O/r05_dp_sim_$tag.txt \
    -v "single:" -v "x0:--simulate-allreduce-us 0" -v "x30:--simulate-allreduce-us 30" -v "x60:--simulate-allreduce-us 60" -v "x100:--simulate-allreduce-us 100" \
    -v "x30_split:--simulate-allreduce-us 30 --dp-split 1" -v "x60_split:--simulate-allreduce-us 60 --dp-split 1" -v "x100_split:--simulate-allreduce-us 100 --dp-split 1" \
    -- $args --steps 200 --warmup 20 2>&1 | tail -9
done
