#!/bin/bash
# Runs every bench mode reported in DESIGN.md §5 on one box and writes the JSON lines + rocprofv3 kernel stats
# to gpurun_out/refresh/ (copy into profiles/ afterwards).   usage: tools/refresh_profiles.sh [rNN]
set -e
r=${1:-r02}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/refresh; mkdir -p $out
cd $root
python bench.py > $out/${r}_bench_c2_default.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${r}_bench_c2_default_k20_w5.json
python bench.py --no-cpu-baseline --no-calibrate --steps 20 --warmup 5 > $out/${r}_bench_c2_default_k20_w5_no_calibrate.json
python bench.py --no-cpu-baseline --no-overlap > $out/${r}_bench_c2_no_overlap.json
python bench.py --no-cpu-baseline --zipf > $out/${r}_bench_c2_zipf.json
python bench.py --no-cpu-baseline --ragged > $out/${r}_bench_c2_ragged.json
python bench.py --no-cpu-baseline --inputs index > $out/${r}_bench_c2_inputs_index.json
python bench.py --no-cpu-baseline --inputs cache > $out/${r}_bench_c2_inputs_cache.json
python bench.py --no-cpu-baseline --k1-only > $out/${r}_bench_c2_k1only.json
python bench.py --no-cpu-baseline --h2d > $out/${r}_bench_c2_h2d.json
python bench.py --no-cpu-baseline --e2e > $out/${r}_bench_c2_e2e_epoch200.json
python bench.py --no-cpu-baseline --e2e --steps 1000 > $out/${r}_bench_c2_e2e_epoch1000.json
python bench.py --no-cpu-baseline --e2e --inputs index > $out/${r}_bench_c2_e2e_index_epoch200.json
python bench.py --no-cpu-baseline --loss inbatch > $out/${r}_bench_c2_inbatch_b2048.json
python bench.py --no-cpu-baseline --loss inbatch --batch 4096 > $out/${r}_bench_inbatch_b4096.json
python bench.py --no-cpu-baseline --config c3 > $out/${r}_bench_c3_share.json
cd /tmp && export TMPDIR=/tmp
for mode in default no_overlap; do
  extra=""; [ $mode = no_overlap ] && extra="--no-overlap"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$mode -o ks -- python3 $root/bench.py --no-cpu-baseline $extra > /dev/null 2>&1
  cp $(find $out/prof_$mode -name "*kernel_stats.csv" | head -1) $out/${r}_bench_c2_${mode}_kernel_stats.csv
  if [ $mode = default ]; then
    python3 $root/tools/timeline.py $(find $out/prof_$mode -name "*kernel_trace.csv" | head -1) 120 2 > $out/${r}_timeline_c2_default.txt || true
  fi
  rm -rf $out/prof_$mode
done
echo refreshed
