# binned K1 (4-wave workgroups) beside different projection tiles: does anything move the step off 5.5 ms?
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r04_binned; mkdir -p $out; cd $root
W="--tune k1_stream_wgs=4"
python tools/ab_tune.py --rounds 2 --out $out/tiles.txt -v "auto:$W" -v "t11:$W --tune proj_tile=11 --tune tn_tile=11" -v "t21:$W --tune proj_tile=21 --tune tn_tile=11" -v "t22bk32:$W --tune proj_tile=22 --tune proj_bk=32" -v "noxcd:$W --tune xcd_order=-1" -v "fork:$W --tune backward=fork" -v "nooverlap:--no-overlap" -- --config c4 --steps 20 --warmup 4 > $out/tiles.log 2>&1; tail -8 $out/tiles.log
