#!/bin/bash
# Round 3: the residency-limited streaming K1 on the BERT-sized (cache-resident) table — does a gather that keeps a fixed,
# small number of waves per CU disturb the projection chain less than the XCD-sliced form?   -> gpurun_out/streamc2/
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/streamc2; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["ms_per_step"]*1e3,1), "us  K1 in-loop", round(r["avg_launch_ms"]*1e3,1), "alone", round(r["alone"]["avg_launch_ms"]*1e3,1))'
rm -f $out/ab.txt
IB="python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096"
C2="python3 bench.py --no-cpu-baseline --no-secondary"
echo "inbatch sliced (default): $($IB | python3 -c "$pick")" >> $out/ab.txt
for w in 1 2 3; do
  echo "inbatch stream wgs=$w: $(TT_K1_STREAM=1 TT_K1_STREAM_WGS=$w $IB | python3 -c "$pick")" >> $out/ab.txt
done
echo "inbatch stream wgs=2 64x64 S: $(TT_K1_STREAM=1 TT_K1_STREAM_WGS=2 TT_INBATCH_TILE=1 $IB | python3 -c "$pick")" >> $out/ab.txt
echo "c2 sliced (default): $($C2 | python3 -c "$pick")" >> $out/ab.txt
for w in 1 2 3 4; do
  echo "c2 stream wgs=$w: $(TT_K1_STREAM=1 TT_K1_STREAM_WGS=$w $C2 | python3 -c "$pick")" >> $out/ab.txt
done
echo "c2 stream wgs=2 TT_TILE=11: $(TT_K1_STREAM=1 TT_K1_STREAM_WGS=2 TT_TILE=11 $C2 | python3 -c "$pick")" >> $out/ab.txt
echo "c2 stream wgs=3 TT_TILE=11: $(TT_K1_STREAM=1 TT_K1_STREAM_WGS=3 TT_TILE=11 $C2 | python3 -c "$pick")" >> $out/ab.txt
cat $out/ab.txt
