#!/bin/bash
# Round 5, step 5: the binned gather's sensitivity to the class-block size; the drop-in route with stream-ordered loss prints.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s5; mkdir -p $out
cd $root
python -m pytest tests/test_gpu_parity.py tests/test_gpu_model_api.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log; tail -3 $out/tests.log
python tools/ab_tune.py --rounds 1 --out $out/binned_vocab.txt -v "v250k:" -v "v187k:--vocab 187500" -v "v125k:--vocab 125000" -v "v62k:--vocab 62500" -- --config c4 --k1-only --tune k1_form=binned --steps 12 --warmup 3 2>&1 | tail -5
python tools/ab_tune.py --rounds 3 --out $out/module_path.txt -v "module:--path module" -v "engine:" -- --steps 200 --warmup 20 2>&1 | tail -3
echo done
