#!/bin/bash
# usage (on the GPU box): tools/pmc_occupancy.sh  — SQ wave-residency counters of every kernel, gathers beside the chain
# (default) vs everything on one stream (--no-overlap): is a kernel slower beside the other stream because its waves
# wait longer (same wave-cycles) or because fewer of them are resident at a time (same cycles per wave, longer launch)?
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in default no_overlap; do
  extra=""; [ $mode = no_overlap ] && extra="--no-overlap"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_occ_$mode -o o -- python3 $root/bench.py --no-cpu-baseline --no-calibrate --steps 40 --warmup 60 $extra > /dev/null 2>&1 || exit 1
done
python3 - $root/gpurun_out <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
for mode in ("default", "no_overlap"):
    f = glob.glob(f"{root}/pmc_occ_{mode}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = ("K1 gather" if "pool_sliced" in n else "K1 combine" if "pool_combine" in n else "gemm NT" if "gemm_f32_kernel<0" in n else
               "gemm NN" if "gemm_f32_kernel<1" in n else "gemm TN" if "gemm_f32_kernel<2" in n else "dW2+dH" if "gemm_tn_nn" in n else
               "triplet" if "triplet_kernel" in n else "adam" if "adam_kernel" in n else None)
        if key:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== {mode}")
    for k, c in acc.items():
        m = {n: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for n, v in c.items()}
        waves, wc, gui = m.get("SQ_WAVES", 0), m.get("SQ_WAVE_CYCLES", 0), m.get("GRBM_GUI_ACTIVE", 0)
        print(f"{k:11s} waves {waves:9.0f}  wave_cycles {wc:13.0f}  cycles/wave {wc / max(waves, 1):9.0f}  kernel_cycles(GRBM/8) {gui / 8:9.0f}  "
              f"avg resident waves/CU {wc * 4 / max(gui / 8, 1) / 256:6.1f}")
PY
