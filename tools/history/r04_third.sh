#!/bin/bash
O=gpurun_out/r04_third; mkdir -p $O
python tools/ab_tune.py --rounds 2 --out $O/c4_wgs3.txt \
  -v "w2r6:" -v "w3r3:--tune k1_stream_wgs=3 --tune k1_stream_rows=3" -v "w3r4:--tune k1_stream_wgs=3 --tune k1_stream_rows=4" \
  -v "w2r8:--tune k1_stream_rows=8" -v "w4r3:--tune k1_stream_wgs=4 --tune k1_stream_rows=3" \
  -- --config c4 --steps 30 --warmup 8 2>&1 | tail -7
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/ks_c2 -o ks -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 20 > $R/$O/ks_c2.json 2>/dev/null
cd $R
python tools/timeline.py $(find $O/ks_c2 -name "*kernel_trace.csv" | head -1) 150 2 > $O/timeline_c2.txt; head -24 $O/timeline_c2.txt; rm -rf $O/ks_c2
bash tools/r04_dp_sim.sh
