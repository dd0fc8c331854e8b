#!/bin/bash
# Round 5 baseline: kernel traces of the data-parallel step (stand-in exchange of 60 us) at the per-GPU shapes of 8 GPUs.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_base; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
prof() { tag=$1; first=$2; shift 2
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o ks -- python3 $root/bench.py --no-cpu-baseline --no-secondary "$@" > $out/${tag}.json 2> $out/${tag}.err
  cp $(find $out/prof_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
  python3 $root/tools/timeline.py $(find $out/prof_$tag -name "*kernel_trace.csv" | head -1) $first 2 > $out/timeline_${tag}.txt || true
  rm -rf $out/prof_$tag; }
prof c3_x60 120 --config c3 --simulate-allreduce-us 60
prof c3_x0 120 --config c3 --simulate-allreduce-us 0
prof c2_x60 120 --config c2 --simulate-allreduce-us 60
prof inbatch512_x60 120 --config c2 --loss inbatch --batch 512 --simulate-allreduce-us 60
prof c3_cache 120 --config c3 --inputs cache
echo done
