# XCD-aware tile order beside the column-sliced gather, across shapes (interleaved A/B per shape)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r04_xcd; mkdir -p $out; cd $root
ab() { tag=$1; shift; python tools/ab_tune.py --rounds 3 --out $out/$tag.txt -v "base:" -v "xcd:--tune xcd_order=1" -- "$@" > $out/$tag.log 2>&1; echo "## $tag: $*"; tail -2 $out/$tag.log; }
ab c2_b1024 --batch 1024 --steps 300 --warmup 30
ab c2_b4096 --batch 4096 --steps 200 --warmup 20
ab c3_b2048 --config c3 --batch 2048 --steps 200 --warmup 20
ab c3_b512 --config c3 --batch 512 --steps 300 --warmup 30
ab inbatch_b4096 --loss inbatch --batch 4096 --steps 200 --warmup 20
ab c2_index --inputs index --steps 300 --warmup 30
