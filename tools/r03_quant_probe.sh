#!/bin/bash
# Round 3: does the 64x64-tile round quantisation (576 tiles on 256 CUs = 2.25 rounds, paid as 3) still cost anything INSIDE
# the step, beside the column-sliced gather?  Batches whose tile counts fall just below a multiple of 256 against the
# metric's 2048:  B = 1792 -> 504 tiles (1.97 rounds), B = 2048 -> 576 (2.25), B = 2688 -> 756 (2.95), B = 2304 -> 648 (2.53).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/quant; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=j["config"]["batch_per_gpu"]; print(b, "triplets:", round(j["ms_per_step"]*1e3,1), "us per step =", round(j["ms_per_step"]*1e6/b,2), "ns per triplet")'
rm -f $out/ab.txt
for rep in 1 2; do for b in 1792 2048 2304 2688; do
echo "overlapped B=$b: $(python3 bench.py --no-cpu-baseline --no-secondary --batch $b | python3 -c "$pick")" >> $out/ab.txt
echo "cache-fed B=$b: $(python3 bench.py --no-cpu-baseline --no-secondary --inputs cache --batch $b | python3 -c "$pick")" >> $out/ab.txt
done; done
sort $out/ab.txt
