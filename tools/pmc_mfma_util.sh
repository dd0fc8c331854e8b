#!/bin/bash
# MFMA utilisation of the projection kernels per kernel template (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over
# GRBM_GUI_ACTIVE / 8 XCDs; counter collection serialises kernels, so every kernel is measured alone):
# configs[1] (default tiles, and with the 96x96 forward tiles TT_G16=1), configs[3].   -> gpurun_out/mfma/r03_k2_mfma_util.txt
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/mfma; mkdir -p $out
rm -f $out/r03_k2_mfma_util.txt
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; label=$2; cfg=$3; steps=$4
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_$tag -o p -- python3 $root/bench.py --no-cpu-baseline --no-secondary --no-calibrate --no-overlap --config $cfg --steps $steps --warmup 2 > /dev/null 2>&1 || return 1
  python3 - $out/pmc_$tag "$label" <<'PY' >> $out/r03_k2_mfma_util.txt
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "gemm" in n:
            mt = re.search(r"(gemm\w*_kernel<[^>]*>)", n)
            acc[mt.group(1) if mt else n[:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("==", sys.argv[2], "(template args: layout 0 NT / 1 NN / 2 TN, epilogue, tile wm, wn, [vector path], [K-step])")
for k, c in sorted(acc.items()):
    m = {n: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for n, v in c.items()}
    print(f"{k:48s} mfma_util {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(m.get('GRBM_GUI_ACTIVE', 0) / 8, 1):5.3f}   launches {len(c.get('GRBM_GUI_ACTIVE', []))}")
PY
  rm -rf $out/pmc_$tag; }
run c2 "configs[1] B=2048 D=P=384 (one stream)" c2 40
TT_G16=1 run c2g16 "configs[1] with the 96x96 forward tiles (TT_G16=1)" c2 40
run c4 "configs[3] B=16384 D=P=768 (one stream: K-step 32 tiles)" c4 6
TT_BK=16 TT_TN_TILE=22 TT_XCD=1 run c4bk16 "configs[3] with the tiles used beside the streaming gather (K-step 16, 128x128 TN, XCD order)" c4 6
cat $out/r03_k2_mfma_util.txt
