#!/bin/bash
# Round 3: streaming K1 (persistent, per-CU registry) beside the projection chain at configs[3].
#   usage (GPU box): tools/r03_c4_stream.sh  -> gpurun_out/c4stream/
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/c4stream; mkdir -p $out
cd $root
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream or projection or metric_size or engine_vs_oracle or awkward" > $out/pytest_stream.txt 2>&1 || { tail -30 $out/pytest_stream.txt; exit 1; }
tail -3 $out/pytest_stream.txt
B="python3 bench.py --config c4 --no-cpu-baseline --steps 12 --warmup 4"
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["ms_per_step"],3), "K1 in-loop ms", round(r["avg_launch_ms"],3), "alone", round(r["alone"]["avg_launch_ms"],3), j["calibration"]["ms_per_step"])'
echo "old form (TT_K1_STREAM=0): $(TT_K1_STREAM=0 $B | python3 -c "$pick")" >> $out/sweep.txt
echo "default (stream, bk16): $($B | python3 -c "$pick")" >> $out/sweep.txt
for wgs in 1 2; do for bk in 16 32; do for u in 0 4; do
  echo "stream wgs=$wgs bk=$bk U=$u: $(TT_K1_STREAM_WGS=$wgs TT_BK=$bk TT_K1_STREAM_U=$u $B | python3 -c "$pick")" >> $out/sweep.txt
done; done; done
echo "stream wgs=1 bk16 U=8: $(TT_K1_STREAM_WGS=1 TT_K1_STREAM_U=8 $B | python3 -c "$pick")" >> $out/sweep.txt
echo "stream wgs=1 bk16 lds80: $(TT_K1_STREAM_LDS_KB=80 $B | python3 -c "$pick")" >> $out/sweep.txt
echo "stream wgs=3 tile=21 U=4: $(TT_K1_STREAM_WGS=3 TT_TILE=21 TT_K1_STREAM_U=4 $B | python3 -c "$pick")" >> $out/sweep.txt
echo "stream wgs=2 tile=21 U=0: $(TT_K1_STREAM_WGS=2 TT_TILE=21 $B | python3 -c "$pick")" >> $out/sweep.txt
echo "no-overlap bk16: $(TT_BK=16 $B --no-overlap | python3 -c "$pick")" >> $out/sweep.txt
cat $out/sweep.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o ks -- python3 $root/bench.py --config c4 --no-cpu-baseline --steps 12 --warmup 4 > /dev/null 2>&1 || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/c4_stream_kernel_stats.csv
python3 $root/tools/timeline.py $(find $out/trace -name "*kernel_trace.csv" | head -1) 8 2 > $out/c4_stream_timeline.txt
rm -rf $out/trace
head -40 $out/c4_stream_timeline.txt
