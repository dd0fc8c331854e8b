#!/bin/bash
# Round 3: column-sliced K1, second pass — 16 loads in flight per wave, 4 or 8 waves per workgroup; and whether the
# run-to-run loss difference of the first pass comes from calibrate()'s choice (it should vanish with --no-calibrate).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/colslice; mkdir -p $out
cd $root
rc=0

[ $rc -ne 0 ] && exit $rc
rm -f $out/k1_alone2.txt
for w in 4 8; do
echo "== column slices, $w waves per workgroup, D=384" >> $out/k1_alone2.txt
TT_K1_COLS=1 TT_K1_COL_WAVES=$w CASES="30522,6144,128;30522,3072,128;30522,6144,64" timeout -k 10 300 python3 tools/k1_sliced_ab.py >> $out/k1_alone2.txt 2>&1 || exit 1
done
grep -E "==|sliced|V=" $out/k1_alone2.txt
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"] or {}; print(round(j["ms_per_step"]*1e3,1), "us  loss", j["avg_loss"], " K1 in-loop", round((r.get("avg_launch_ms") or 0)*1e3,1), " alone", round(((r.get("alone") or {}).get("avg_launch_ms") or 0)*1e3,1))'
rm -f $out/ab2.txt
for rep in 1 2; do
for v in "TT_K1_COLS=0" "TT_K1_COLS=1 TT_K1_COL_WAVES=4" "TT_K1_COLS=1 TT_K1_COL_WAVES=8"; do
echo "c2 default $v: $(env $v python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab2.txt
echo "c2 default no-calibrate $v: $(env $v python3 bench.py --no-cpu-baseline --no-secondary --no-calibrate | python3 -c "$pick")" >> $out/ab2.txt
echo "c2 one stream $v: $(env $v python3 bench.py --no-cpu-baseline --no-secondary --no-overlap | python3 -c "$pick")" >> $out/ab2.txt
done; done
sort $out/ab2.txt
