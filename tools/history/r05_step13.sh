#!/bin/bash
# Round 5, step 13: token-store batches staged by a copy pass on the third stream (no row indirection in the gather).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s13; mkdir -p $out
cd $root
python -m pytest tests/test_gpu_parity.py tests/test_gpu_model_api.py tests/test_gpu_tokenizer.py tests/test_gpu_dp.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do
for v in true false; do
python bench.py --no-cpu-baseline --no-secondary --e2e --steps 1000 --trainer pack_roles=$v > $out/e2e_$v.json 2>/dev/null
python -c "
import json; j=json.loads(open('$out/e2e_$v.json').read().strip().splitlines()[-1]); print('staged=$v e2e rows3', round(j['value']/1e6,3), 'M/s', round(j['ms_per_step']*1e3,1), 'us/step')"
python bench.py --no-cpu-baseline --no-secondary --e2e --steps 1000 --inputs index --trainer pack_roles=$v > $out/e2e_idx_$v.json 2>/dev/null
python -c "
import json; j=json.loads(open('$out/e2e_idx_$v.json').read().strip().splitlines()[-1]); print('staged=$v e2e rows2', round(j['value']/1e6,3), 'M/s', round(j['ms_per_step']*1e3,1), 'us/step')"
done; done
