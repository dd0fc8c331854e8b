root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r04_binned; mkdir -p $out; cd $root
python tools/ab_tune.py --rounds 2 --out $out/depth.txt -v "d2:" -v "d4:--tune proj_depth=4" -v "d6:--tune proj_depth=6" -- --config c4 --steps 20 --warmup 4 > $out/depth.log 2>&1; tail -4 $out/depth.log
