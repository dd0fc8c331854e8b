root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r04_binned; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o ks -- python3 $root/bench.py --no-cpu-baseline --no-secondary --config c4 --steps 20 --warmup 4 > $out/bench_c4.json 2>/dev/null
python3 $root/tools/timeline.py $(find $out/prof -name "*kernel_trace.csv" | head -1) 12 2 > $out/timeline_c4.txt
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats_c4.csv
rm -rf $out/prof
cat $out/timeline_c4.txt
