#!/bin/bash
# round 4, second pass: module path after the launch cuts; streaming-K1 admission A/B at configs[3] (interleaved);
# kernel timeline of configs[2]'s one-GPU share (what 8 GPUs run per rank).
set -o pipefail
O=gpurun_out/r04_second; mkdir -p $O
python -m pytest tests/test_gpu_model_api.py tests/test_gpu_parity.py -x -q -m gpu \
  -k "adam_multi or cosine_distance or pretokenised or accumulation or optimized or module_path or golden_edges" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python bench.py --path module --steps 200 --warmup 20 > $O/c2_module.json 2> $O/c2_module.err; python -c "import json;r=json.load(open('$O/c2_module.json'));print('module path %.2f M/s %.1f us'%(r['value']/1e6,r['ms_per_step']*1e3))"
python tools/ab_tune.py --rounds 3 --out $O/c4_admission_ab.txt -v "offer2:--tune k1_stream_offer=2" -v "offer16:" \
   -v "o16_linger8:--tune k1_stream_linger=8" -v "o16_linger32:--tune k1_stream_linger=32" -v "o4_linger16:--tune k1_stream_offer=4 --tune k1_stream_linger=16" \
   -- --config c4 --steps 30 --warmup 8 2>&1 | tail -8
# kernel trace of the c3 share
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/ks_c3 -o ks -- python3 $R/bench.py --no-cpu-baseline --no-secondary --config c3 --steps 200 --warmup 20 > $R/$O/ks_c3.json 2>/dev/null
cd $R
T=$(find $O/ks_c3 -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T 150 2 > $O/timeline_c3.txt; head -40 $O/timeline_c3.txt
S=$(find $O/ks_c3 -name "*kernel_stats.csv" | head -1); cp $S $O/c3_kernel_stats.csv; rm -rf $O/ks_c3
