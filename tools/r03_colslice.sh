#!/bin/bash
# Round 3: column-sliced K1 (32-column slices: one 128-byte line per row, a slice fits an XCD's L2) against the
# row-sliced form (TT_K1_COLS=0) — parity, K1 alone, then the configs[1] step.   -> gpurun_out/colslice/
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/colslice; mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or metric_size" > $out/pytest.txt 2>&1; rc=$?
tail -5 $out/pytest.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
for c in 0 1; do
echo "== TT_K1_COLS=$c D=384" >> $out/k1_alone.txt
TT_K1_COLS=$c CASES="30522,6144,128;20000,6144,128;10000,6144,128;30522,3072,128;30522,6144,64;30522,12288,128" timeout -k 10 300 python3 tools/k1_sliced_ab.py >> $out/k1_alone.txt 2>&1 || exit 1
echo "== TT_K1_COLS=$c D=768" >> $out/k1_alone.txt
D=768 TT_K1_COLS=$c CASES="30522,6144,128;30522,3072,256" timeout -k 10 300 python3 tools/k1_sliced_ab.py >> $out/k1_alone.txt 2>&1 || exit 1
done
grep -E "==|sliced|V=" $out/k1_alone.txt
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"] or {}; print(round(j["ms_per_step"]*1e3,1), "us  loss", j["avg_loss"], " K1 in-loop", round((r.get("avg_launch_ms") or 0)*1e3,1), " alone", round(((r.get("alone") or {}).get("avg_launch_ms") or 0)*1e3,1))'
rm -f $out/ab.txt
for rep in 1 2; do for c in 0 1; do
echo "c2 default TT_K1_COLS=$c: $(TT_K1_COLS=$c python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab.txt
echo "c2 one stream TT_K1_COLS=$c: $(TT_K1_COLS=$c python3 bench.py --no-cpu-baseline --no-secondary --no-overlap | python3 -c "$pick")" >> $out/ab.txt
echo "c2 index TT_K1_COLS=$c: $(TT_K1_COLS=$c python3 bench.py --no-cpu-baseline --no-secondary --inputs index | python3 -c "$pick")" >> $out/ab.txt
echo "c3 share TT_K1_COLS=$c: $(TT_K1_COLS=$c python3 bench.py --no-cpu-baseline --no-secondary --config c3 | python3 -c "$pick")" >> $out/ab.txt
echo "inbatch b4096 TT_K1_COLS=$c: $(TT_K1_COLS=$c python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096 | python3 -c "$pick")" >> $out/ab.txt
done; done
sort $out/ab.txt
