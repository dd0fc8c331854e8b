#!/bin/bash
# Round 6, SURVEY §8 f2: bench.py --eval on one box — the JSON line, rocprofv3 kernel stats of the same command, and the
# MFMA utilisation of the cosine-score GEMM (--pmc pass of its own).   usage: bash tools/r06_eval_prof.sh [out_dir]
root=$GRAFT_REPO_ROOT; out=${1:-$root/gpurun_out/r06_eval}; mkdir -p $out
cd $root && python bench.py --eval > $out/r06_eval.json 2> $out/r06_eval.log || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 $root/bench.py --eval --no-cpu-baseline > /dev/null 2>&1 || exit 1
cp $(find $out/ks -name "*kernel_stats.csv" | head -1) $out/r06_eval_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc -o p -- python3 $root/bench.py --eval --no-cpu-baseline > /dev/null 2>&1 || exit 1
python3 - $out/pmc <<'PY' > $out/r06_eval_mfma_util.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "gemm" in n or "ndcg" in n or "unit_rows" in n:
            key = (n[:90], row.get("Grid_Size", ""))
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("bench.py --eval under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (mfma_util = MFMA_BUSY / 1024 / (GUI_ACTIVE / 8), per launch, averaged)")
for (k, grid), c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(f"{k:92s} grid {grid:>9s} launches {len(c.get('GRBM_GUI_ACTIVE', [])):4d}  gui_active {m.get('GRBM_GUI_ACTIVE', 0):10.0f}  mfma_util {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(m.get('GRBM_GUI_ACTIVE', 0) / 8, 1):5.3f}")
PY
rm -rf $out/ks $out/pmc
cat $out/r06_eval_mfma_util.txt
