# binned K1 beside the chain: rows in flight per lane group (U) x waves per workgroup (experiment knobs)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r04_binned; mkdir -p $out; cd $root
python tools/ab_tune.py --rounds 2 --out $out/occ.txt -v "w8u8:" -v "w4u8:--tune k1_stream_wgs=4" -v "w8u4:--tune k1_stream_rows=4" -v "w4u4:--tune k1_stream_wgs=4 --tune k1_stream_rows=4" -v "w4u2:--tune k1_stream_wgs=4 --tune k1_stream_rows=2" -- --config c4 --steps 20 --warmup 4 > $out/occ.log 2>&1; tail -6 $out/occ.log
