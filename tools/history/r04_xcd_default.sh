# the new default (XCD-aware order from 6144 rows beside the sliced gather) against off, per shape
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r04_xcd; mkdir -p $out; cd $root
ab() { tag=$1; shift; python tools/ab_tune.py --rounds 3 --out $out/$tag.txt -v "off:--tune xcd_order=-1" -v "auto:" -- "$@" > $out/$tag.log 2>&1; echo "## $tag: $*"; tail -2 $out/$tag.log; }
ab d_c2 --steps 300 --warmup 30
ab d_c2_cache --inputs cache --steps 300 --warmup 30
ab d_c2_sim60 --simulate-allreduce-us 60 --steps 300 --warmup 30
ab d_c2_nooverlap --no-overlap --steps 200 --warmup 20
