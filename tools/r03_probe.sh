#!/bin/bash
# Round 3 probe at configs[1]: 96x96 forward tiles (gemm16, NT only) FORCED, where the chain has the chip to itself.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/probe2; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"]*1e3,1), "us")'
rm -f $out/g16.txt
for rep in 1 2 3; do
echo "cache 64x64: $(python3 bench.py --no-cpu-baseline --no-secondary --inputs cache | python3 -c "$pick")" >> $out/g16.txt
echo "cache 96x96 forward (forced): $(TT_G16=1 TT_G16_FORCE=1 python3 bench.py --no-cpu-baseline --no-secondary --inputs cache | python3 -c "$pick")" >> $out/g16.txt
echo "one stream 64x64: $(python3 bench.py --no-cpu-baseline --no-secondary --no-overlap | python3 -c "$pick")" >> $out/g16.txt
echo "one stream 96x96 forward (forced): $(TT_G16=1 TT_G16_FORCE=1 python3 bench.py --no-cpu-baseline --no-secondary --no-overlap | python3 -c "$pick")" >> $out/g16.txt
done
cat $out/g16.txt
