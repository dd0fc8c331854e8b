#!/bin/bash
# Round 5, step 7: the drop-in route with the packing pass on a third stream; configs[3] under skewed / ragged ids.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s7; mkdir -p $out
cd $root
python -m pytest tests/test_gpu_parity.py tests/test_gpu_model_api.py tests/test_gpu_tokenizer.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log; tail -3 $out/tests.log
python tools/ab_tune.py --rounds 3 --out $out/module_path.txt -v "module:--path module" -v "engine:" -- --steps 200 --warmup 20 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-secondary"
$B --config c4 --steps 30 --warmup 6 --zipf > $out/c4_zipf.json
$B --config c4 --steps 30 --warmup 6 --ragged > $out/c4_ragged.json
$B --config c4 --steps 30 --warmup 6 --zipf --tune k1_form=stream > $out/c4_zipf_stream.json
$B --config c4 --steps 30 --warmup 6 --ragged --tune k1_form=stream > $out/c4_ragged_stream.json
for f in c4_zipf c4_ragged c4_zipf_stream c4_ragged_stream; do python -c "
import json; j=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); r=j['roofline']; print('$f', round(j['ms_per_step'],3), 'ms/step  K1 in loop', round(r['avg_launch_ms'],3), 'alone', round(r['alone']['avg_launch_ms'],3), r['kernel'][:40], 'frac', round(r['frac'],3), r['bound'])"; done
echo done
