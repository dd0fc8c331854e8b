#!/bin/bash
# Round 3: configs[3] with the K-step-16 tiles compiled into 128 registers (second build of the library,
# -DTT_BK16_WAVES=4) so that two tiles AND two gather workgroups of 7 rows (128 registers) fit a CU.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/c4w4; mkdir -p $out
cd $root
W4=$root/two-towers-overlords_amd/libtt_hotpath_w4.so
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["ms_per_step"],3), "K1 in-loop ms", round(r["avg_launch_ms"],3), "alone", round(r["alone"]["avg_launch_ms"],3), j["calibration"]["ms_per_step"])'
B="python3 bench.py --config c4 --no-cpu-baseline --steps 30 --warmup 8"
rm -f $out/ab.txt
for rep in 1 2; do
echo "w3 (default build) U=6: $($B | python3 -c "$pick")" >> $out/ab.txt
echo "w4 U=6: $(TT_HOTPATH_LIB=$W4 $B | python3 -c "$pick")" >> $out/ab.txt
echo "w4 U=7: $(TT_HOTPATH_LIB=$W4 TT_K1_STREAM_U=7 $B | python3 -c "$pick")" >> $out/ab.txt
echo "w4 U=8: $(TT_HOTPATH_LIB=$W4 TT_K1_STREAM_U=8 $B | python3 -c "$pick")" >> $out/ab.txt
done
echo "w4 U=7 no-overlap: $(TT_HOTPATH_LIB=$W4 TT_K1_STREAM_U=7 $B --no-overlap | python3 -c "$pick")" >> $out/ab.txt
echo "w4 wgs=3 U=4: $(TT_HOTPATH_LIB=$W4 TT_K1_STREAM_WGS=3 TT_K1_STREAM_U=4 $B | python3 -c "$pick")" >> $out/ab.txt
cat $out/ab.txt
