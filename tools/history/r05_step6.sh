#!/bin/bash
# Round 5, step 6: binned K1 with one launch per slice (parity, K1 alone, the configs[3] step); the drop-in route with the
# packing pass on a third stream.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s6; mkdir -p $out
cd $root
python -m pytest tests/test_gpu_parity.py tests/test_gpu_model_api.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log; tail -3 $out/tests.log
python tools/ab_tune.py --rounds 2 --out $out/binned_k1only.txt -v "binned:" -- --config c4 --k1-only --tune k1_form=binned --steps 12 --warmup 3 2>&1 | tail -2
python tools/ab_tune.py --rounds 2 --out $out/c4_step.txt -v "c4:" -- --config c4 --steps 20 --warmup 4 2>&1 | tail -2
python tools/ab_tune.py --rounds 3 --out $out/module_path.txt -v "module:--path module" -v "engine:" -- --steps 200 --warmup 20 2>&1 | tail -3
echo done
