#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s10; mkdir -p $out
cd $root
python -m pytest tests/test_gpu_parity.py tests/test_gpu_model_api.py tests/test_gpu_tokenizer.py tests/test_gpu_dp.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do
for v in 0 1; do
TT_AB_STORE_LENS=$v python bench.py --no-cpu-baseline --no-secondary --e2e --steps 1000 > $out/e2e_lens$v.json 2>/dev/null
python -c "
import json; j=json.loads(open('$out/e2e_lens$v.json').read().strip().splitlines()[-1]); print('store_lens=$v e2e 1000 batches', round(j['value']/1e6,3), 'M triplets/s', round(j['ms_per_step']*1e3,1), 'us/step')"
TT_AB_STORE_LENS=$v python bench.py --no-cpu-baseline --no-secondary --e2e --steps 1000 --inputs index > $out/e2e_idx_lens$v.json 2>/dev/null
python -c "
import json; j=json.loads(open('$out/e2e_idx_lens$v.json').read().strip().splitlines()[-1]); print('store_lens=$v e2e rows2/index', round(j['value']/1e6,3), 'M triplets/s', round(j['ms_per_step']*1e3,1), 'us/step')"
done; done
