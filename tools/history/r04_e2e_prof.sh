root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/e2e_prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o ks -- python3 $root/bench.py --no-cpu-baseline --no-secondary --e2e --steps 300 > $out/bench.json 2>/dev/null
python3 $root/tools/timeline.py $(find $out/prof -name "*kernel_trace.csv" | head -1) 150 2 > $out/timeline.txt
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/prof
cat $out/timeline.txt
