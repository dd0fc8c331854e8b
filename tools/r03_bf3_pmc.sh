#!/bin/bash
# PMC view of the split-bf16 kernels against the fp32 ones, configs[3] one stream (kernels alone)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/bf3pmc; mkdir -p $out
rm -f $out/pmc.txt
cd /tmp && export TMPDIR=/tmp
for bf in 0 1; do
  TT_BF3=$bf rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/p$bf -o p -- python3 $root/bench.py --no-cpu-baseline --no-calibrate --no-overlap --config c4 --steps 4 --warmup 2 > /dev/null 2>&1 || exit 1
  python3 - $out/p$bf $bf <<'PY' >> $out/pmc.txt
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "gemm" in n:
            mt = re.search(r"(gemm\w*_kernel<[^>]*>)", n)
            acc[mt.group(1) if mt else n[:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== TT_BF3=" + sys.argv[2])
for k, c in sorted(acc.items()):
    m = {n: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for n, v in c.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8
    wc = max(m.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{k:40s} kernel_cycles {gui:9.0f} mfma_busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/max(gui,1):5.3f} valu_active/wave_cycles {m.get('SQ_ACTIVE_INST_VALU',0)/wc:5.3f} "
          f"wait_any {m.get('SQ_WAIT_ANY',0)/wc:5.3f} wait_inst_any {m.get('SQ_WAIT_INST_ANY',0)/wc:5.3f} valu_insts {m.get('SQ_INSTS_VALU',0):.3g} resident waves/CU {wc*4/max(gui,1)/256:4.1f}")
PY
  rm -rf $out/p$bf
done
cat $out/pmc.txt
