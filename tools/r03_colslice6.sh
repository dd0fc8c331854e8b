#!/bin/bash
# Round 3: column-sliced K1 — the K1 parity tests, fabric-side traffic per launch (PMC, two passes), and the two
# slicings alone over vocabulary sizes around the L2 boundary.   -> gpurun_out/colslice/
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/colslice; mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $out/pytest6.txt 2>&1; rc=$?
tail -3 $out/pytest6.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
bash tools/pmc_k1.sh c2 1220542464 pool_colsliced_kernel,pool_colnorm_kernel > $out/pmc.txt 2>&1 || { tail -5 $out/pmc.txt; exit 1; }
cp gpurun_out/k1_traffic_c2.json $out/r03_k1_traffic_c2.json
cat $out/r03_k1_traffic_c2.json
rm -f $out/vsweep.txt
for c in 0 1; do
echo "== TT_K1_COLS=$c" >> $out/vsweep.txt
TT_K1_COLS=$c CASES="30522,6144,128;36000,6144,128;42000,6144,128;50000,6144,128;65000,6144,128;100000,6144,128" timeout -k 10 300 python3 tools/k1_sliced_ab.py >> $out/vsweep.txt 2>&1 || exit 1
done
grep -E "==|sliced|V=" $out/vsweep.txt
