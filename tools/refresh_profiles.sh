#!/bin/bash
# Runs every bench mode reported in DESIGN.md §5 on one box and writes the JSON lines + rocprofv3 kernel stats
# to gpurun_out/refresh/ (copy into profiles/ afterwards).
set -e
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/refresh; mkdir -p $out
cd $root
python bench.py > $out/r01_bench_c2_default.json
python bench.py --no-cpu-baseline --no-overlap > $out/r01_bench_c2_no_overlap.json
python bench.py --no-cpu-baseline --zipf > $out/r01_bench_c2_zipf.json
python bench.py --no-cpu-baseline --ragged > $out/r01_bench_c2_ragged.json
python bench.py --no-cpu-baseline --inputs index > $out/r01_bench_c2_inputs_index.json
python bench.py --no-cpu-baseline --inputs cache > $out/r01_bench_c2_inputs_cache.json
python bench.py --no-cpu-baseline --k1-only > $out/r01_bench_c2_k1only.json
python bench.py --no-cpu-baseline --loss inbatch > $out/r01_bench_c2_inbatch_b2048.json
python bench.py --no-cpu-baseline --loss inbatch --batch 4096 > $out/r01_bench_inbatch_b4096.json
cd /tmp && export TMPDIR=/tmp
for mode in default no_overlap; do
  extra=""; [ $mode = no_overlap ] && extra="--no-overlap"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$mode -o ks -- python3 $root/bench.py --no-cpu-baseline $extra > /dev/null 2>&1
  cp $(find $out/prof_$mode -name "*kernel_stats.csv" | head -1) $out/r01_bench_c2_${mode}_kernel_stats.csv
  rm -rf $out/prof_$mode
done
echo refreshed
