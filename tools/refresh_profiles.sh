#!/bin/bash
# Runs the bench modes reported in DESIGN.md §5 on one box and writes the JSON lines + rocprofv3 kernel stats
# to gpurun_out/refresh/ (copy into profiles/ afterwards).   usage: tools/refresh_profiles.sh [rNN]
set -e
r=${1:-r03}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/refresh; mkdir -p $out
cd $root
python bench.py > $out/${r}_bench_c2_default.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $out/${r}_bench_c2_default_k20_w5.json
python bench.py --no-cpu-baseline --no-secondary --no-overlap > $out/${r}_bench_c2_no_overlap.json
python bench.py --no-cpu-baseline --no-secondary --ragged > $out/${r}_bench_c2_ragged.json
python bench.py --no-cpu-baseline --no-secondary --inputs index > $out/${r}_bench_c2_inputs_index.json
python bench.py --no-cpu-baseline --no-secondary --inputs cache > $out/${r}_bench_c2_inputs_cache.json
python bench.py --no-cpu-baseline --no-secondary --e2e --steps 1000 > $out/${r}_bench_c2_e2e_epoch1000.json
python bench.py --no-cpu-baseline --no-secondary --loss inbatch > $out/${r}_bench_c2_inbatch_b2048.json
python bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096 > $out/${r}_bench_inbatch_b4096.json
python bench.py --no-cpu-baseline --config c3 > $out/${r}_bench_c3_share.json
python bench.py --no-cpu-baseline --config c4 --steps 40 --warmup 8 > $out/${r}_bench_c4_full.json
python bench.py --no-cpu-baseline --config c4 --steps 20 --warmup 4 --no-overlap > $out/${r}_bench_c4_no_overlap.json
python bench.py --no-cpu-baseline --config c4 --steps 12 --warmup 4 --inputs index > $out/${r}_bench_c4_inputs_index.json
TT_K1_STREAM=0 TT_BK=32 TT_XCD=0 TT_TN_TILE=11 python bench.py --no-cpu-baseline --config c4 --steps 20 --warmup 4 > $out/${r}_bench_c4_round2_kernels.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2 -o ks -- python3 $root/bench.py --no-cpu-baseline --no-secondary > /dev/null 2>&1
cp $(find $out/prof_c2 -name "*kernel_stats.csv" | head -1) $out/${r}_bench_c2_default_kernel_stats.csv
python3 $root/tools/timeline.py $(find $out/prof_c2 -name "*kernel_trace.csv" | head -1) 120 2 > $out/${r}_timeline_c2_default.txt || true
rm -rf $out/prof_c2
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c4 -o ks -- python3 $root/bench.py --no-cpu-baseline --config c4 --steps 20 --warmup 4 > /dev/null 2>&1
cp $(find $out/prof_c4 -name "*kernel_stats.csv" | head -1) $out/${r}_bench_c4_full_kernel_stats.csv
python3 $root/tools/timeline.py $(find $out/prof_c4 -name "*kernel_trace.csv" | head -1) 12 2 > $out/${r}_timeline_c4_full.txt || true
rm -rf $out/prof_c4
# MFMA utilisation of the projection kernels, configs[1] and configs[3] (one-stream runs: counter collection serialises kernels anyway)
for cfg in c2 c4; do
  steps=40; [ $cfg = c4 ] && steps=6
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_$cfg -o p -- python3 $root/bench.py --no-cpu-baseline --no-secondary --no-calibrate --config $cfg --steps $steps --warmup 2 > /dev/null 2>&1
  python3 - $out/pmc_$cfg $cfg <<'PY' >> $out/${r}_k2_mfma_util.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "gemm" in n:
            import re
            m = re.search(r"(gemm\w+_kernel<[^>]*>)", n)
            acc[m.group(1) if m else n[:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("==", sys.argv[2], "(SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), second half of the launches)")
for k, c in acc.items():
    m = {n: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for n, v in c.items()}
    print(f"{k:62s} mfma_util {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(m.get('GRBM_GUI_ACTIVE', 0) / 8, 1):5.3f}")
PY
  rm -rf $out/pmc_$cfg
done
cat $out/${r}_k2_mfma_util.txt
echo refreshed
