set -e
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r04_join; mkdir -p $out
cd $root
python tools/ab_tune.py --rounds 3 --out $out/c3_sim60.txt -v "own_wait:--join-carries-gather 0" -v "on_join:--join-carries-gather 1" -- --config c3 --simulate-allreduce-us 60 --steps 300 --warmup 30 > $out/c3_sim60.log 2>&1
python tools/ab_tune.py --rounds 3 --out $out/c2_sim60.txt -v "own_wait:--join-carries-gather 0" -v "on_join:--join-carries-gather 1" -- --simulate-allreduce-us 60 --steps 300 --warmup 30 > $out/c2_sim60.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2 -o ks -- python3 $root/bench.py --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 $root/tools/timeline.py $(find $out/prof_c2 -name "*kernel_trace.csv" | head -1) 120 2 > $out/timeline_c2.txt
rm -rf $out/prof_c2
tail -3 $out/c3_sim60.log; tail -3 $out/c2_sim60.log; cat $out/timeline_c2.txt
