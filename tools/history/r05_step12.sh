#!/bin/bash
# Round 5, step 12: half tiles with K-step 64 split between wave pairs: stamps, parity, A/B in the chain-alone regimes.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s12; mkdir -p $out
cd $root
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTT_GEMM_STAMPS -Iinclude -Itwo-towers-overlords_amd/csrc -Wno-unused-function tools/gemm_stamps.hip -o /tmp/gemm_stamps && timeout -k 10 120 /tmp/gemm_stamps > $out/gemm_stamps.txt 2>&1; grep -E "==|balanced|probe|NT layer" $out/gemm_stamps.txt | cut -c1-330
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "balanced or projection" 2>&1 | tail -3
T="--tune proj_balance=1"
python tools/ab_tune.py --rounds 2 --out $out/bal_c3_x60.txt -v "base:" -v "bal:$T" -v "bal_fork:$T --tune backward=fork" -- --config c3 --simulate-allreduce-us 60 --steps 200 --warmup 20 2>&1 | tail -4
python tools/ab_tune.py --rounds 2 --out $out/bal_c2_cache.txt -v "base:" -v "bal:$T" -- --config c2 --inputs cache --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/bal_c2.txt -v "base:" -v "bal:$T" -- --config c2 --steps 200 --warmup 20 2>&1 | tail -3
