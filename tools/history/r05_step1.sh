#!/bin/bash
# Round 5, step 1: GPU suite, the opcode coexistence probe, the data-parallel stand-in tables with the exchange-regime defaults.
root=$GRAFT_REPO_ROOT; O=$root/gpurun_out/r05_s1; mkdir -p $O
cd $root
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -3 $O/tests.log
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/coexist_probe.hip -o /tmp/coexist_probe && timeout -k 10 120 /tmp/coexist_probe > $O/coexist_probe.txt 2>&1; echo "probe rc=$?"
for spec in "c3:--config c3" "c2:--config c2"; do
  tag=${spec%%:*}; args=${spec#*:}
  python tools/ab_tune.py --rounds 2 --out $O/dp_sim_$tag.txt \
    -v "single:" -v "x0:--simulate-allreduce-us 0" -v "x30:--simulate-allreduce-us 30" -v "x60:--simulate-allreduce-us 60" -v "x100:--simulate-allreduce-us 100" \
    -v "x60_fork:--simulate-allreduce-us 60 --tune backward=fork" \
    -- $args --steps 200 --warmup 20 2>&1 | tail -7
done
