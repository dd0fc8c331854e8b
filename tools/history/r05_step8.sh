#!/bin/bash
# Round 5, step 8: the binned gather's class hash under uniform and skewed ids; the drop-in route's timeline.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s8; mkdir -p $out
cd $root
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "binned or engine" > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log; tail -3 $out/tests.log
python tools/ab_tune.py --rounds 2 --out $out/binned_k1only.txt -v "uniform:" -v "zipf:--zipf" -v "zipf_stream:--zipf --tune k1_form=stream" -- --config c4 --k1-only --tune k1_form=binned --steps 12 --warmup 3 2>&1 | tail -4
python tools/ab_tune.py --rounds 2 --out $out/c4_step.txt -v "uniform:" -v "zipf:--zipf" -v "zipf_stream:--zipf --tune k1_form=stream" -v "zipf_cal:--zipf --calibrate-k1" -v "ragged:--ragged" -- --config c4 --steps 20 --warmup 4 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_mod -o ks -- python3 $root/bench.py --no-cpu-baseline --no-secondary --path module --steps 100 > $out/module.json 2> $out/module.err
cp $(find $out/prof_mod -name "*kernel_stats.csv" | head -1) $out/module_kernel_stats.csv
python3 $root/tools/timeline.py $(find $out/prof_mod -name "*kernel_trace.csv" | head -1) 60 2 > $out/timeline_module.txt
rm -rf $out/prof_mod
echo done
