#!/bin/bash
# Round 5, step 2: what the data-parallel step looks like with either backward schedule under a 60 us stand-in exchange.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s2; mkdir -p $out
cd $root
python -m pytest tests/test_gpu_dp.py tests/test_gpu_model_api.py tests/test_bench_launch.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log; tail -3 $out/tests.log
cd /tmp && export TMPDIR=/tmp
prof() { tag=$1; first=$2; shift 2
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o ks -- python3 $root/bench.py --no-cpu-baseline --no-secondary "$@" > $out/${tag}.json 2> $out/${tag}.err
  cp $(find $out/prof_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
  python3 $root/tools/timeline.py $(find $out/prof_$tag -name "*kernel_trace.csv" | head -1) $first 2 > $out/timeline_${tag}.txt || true
  rm -rf $out/prof_$tag; }
prof c3_x60_auto 120 --config c3 --simulate-allreduce-us 60
prof c3_x60_fused 120 --config c3 --simulate-allreduce-us 60 --tune backward=fused
prof c3_x60_fork 120 --config c3 --simulate-allreduce-us 60 --tune backward=fork
cd $root
python bench.py --no-cpu-baseline --no-secondary --config c3 --simulate-allreduce-us 60 > $out/c3_x60_plain.json 2> $out/c3_x60_plain.err
python bench.py --no-cpu-baseline --no-secondary --config c3 --simulate-allreduce-us 60 --tune backward=fused > $out/c3_x60_fused_plain.json 2>> $out/c3_x60_plain.err
echo done
