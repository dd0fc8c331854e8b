#!/bin/bash
# Round 5, step 4: K-step 64 on the 64x64 tiles (stamps, parity, A/B); the binned gather's sensitivity to the class-block size.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s4; mkdir -p $out
cd $root
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "projection_fwd_bwd or engine_step or engine_awkward or golden" > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log; tail -3 $out/tests.log
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTT_GEMM_STAMPS -Iinclude -Itwo-towers-overlords_amd/csrc -Wno-unused-function tools/gemm_stamps.hip -o /tmp/gemm_stamps && timeout -k 10 120 /tmp/gemm_stamps > $out/gemm_stamps.txt 2>&1; echo "stamps rc=$?"
python tools/ab_tune.py --rounds 2 --out $out/bk64_c3_x60.txt -v "base:" -v "bk64:--tune proj_bk=64" -- --config c3 --simulate-allreduce-us 60 --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/bk64_c3.txt -v "base:" -v "bk64:--tune proj_bk=64" -- --config c3 --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/bk64_c2.txt -v "base:" -v "bk64:--tune proj_bk=64" -- --config c2 --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 2 --out $out/bk64_c2_cache.txt -v "base:" -v "bk64:--tune proj_bk=64" -- --config c2 --inputs cache --steps 200 --warmup 20 2>&1 | tail -3
python tools/ab_tune.py --rounds 1 --out $out/binned_vocab.txt -v "v250k:" -v "v187k:--vocab 187500" -v "v125k:--vocab 125000" -v "v62k_binned:--vocab 62500" -- --config c4 --k1-only --tune k1_form=binned --steps 12 --warmup 3 2>&1 | tail -5
echo done
