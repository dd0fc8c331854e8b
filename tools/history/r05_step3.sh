#!/bin/bash
# Round 5, step 3: pool_roles parity + the drop-in route's step time; where a small GEMM's time goes (phase stamps).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s3; mkdir -p $out
cd $root
python -m pytest tests/test_gpu_parity.py tests/test_gpu_model_api.py -m gpu -x -q -k "pool_roles or train_epoch or engine_route or adam_multi or loss" > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log; tail -3 $out/tests.log
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTT_GEMM_STAMPS -Iinclude -Itwo-towers-overlords_amd/csrc -Wno-unused-function tools/gemm_stamps.hip -o /tmp/gemm_stamps && timeout -k 10 120 /tmp/gemm_stamps > $out/gemm_stamps.txt 2>&1; echo "stamps rc=$?"
python tools/ab_tune.py --rounds 3 --out $out/module_path.txt -v "module:--path module" -v "engine:" -- --steps 200 --warmup 20 2>&1 | tail -3
echo done
