#!/bin/bash
# Round 3: in-batch loss — 64x64 (TT_INBATCH_TILE=1, the default since) vs 128x128 (=2) tiles for the similarity product,
# with and without the gathers beside it.  (profiles/r03_inbatch_tile_ab.txt was taken when 0 meant 128x128.)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/inbatch_tile; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"]*1e3,1), "us")'
rm -f $out/ab.txt
for rep in 1 2; do for t in 2 1; do for b in 4096 2048 8192; do
B="python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch $b"
echo "B=$b overlapped TT_INBATCH_TILE=$t: $(TT_INBATCH_TILE=$t $B | python3 -c "$pick")" >> $out/ab.txt
echo "B=$b one stream TT_INBATCH_TILE=$t: $(TT_INBATCH_TILE=$t $B --no-overlap | python3 -c "$pick")" >> $out/ab.txt
done; done; done
sort $out/ab.txt
