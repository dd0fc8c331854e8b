# K1 with unconditional 32-bit-offset row loads (new build) against the previous build (libtt_hotpath_prev.so), same box
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r04_k1bf; mkdir -p $out; cd $root
P="--lib two-towers-overlords_amd/libtt_hotpath_prev.so"
ab() { tag=$1; shift; python tools/ab_tune.py --rounds 3 --out $out/$tag.txt -v "prev:$P" -v "new:" -- "$@" > $out/$tag.log 2>&1; echo "## $tag: $*"; tail -2 $out/$tag.log; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or metric or engine_matches or bins or awkward" > $out/tests.log 2>&1; tail -2 $out/tests.log
ab c2 --steps 300 --warmup 30
ab c3 --config c3 --steps 300 --warmup 30
ab c2_ragged --ragged --steps 300 --warmup 30
ab c4 --config c4 --steps 20 --warmup 4
