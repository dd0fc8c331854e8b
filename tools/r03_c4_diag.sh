#!/bin/bash
# Round-3 diagnostics of configs[3] (V=250k, D=P=768, S=256, B=16384): how K1 and the projection chain share a CU.
#   usage (on the GPU box): tools/r03_c4_diag.sh   -> gpurun_out/c4diag/
# 1. residency sweep: K1 reservation (workgroups per CU) x projection tile, overlap on; 2. kernel timeline of the
# default; 3. per-kernel PMC (counter collection serialises kernels: each kernel ALONE): L2 hit rate, fabric-side
# reads/writes, MFMA busy.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/c4diag; mkdir -p $out
cd $root
B="python3 bench.py --config c4 --no-cpu-baseline --steps 12 --warmup 4"
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["ms_per_step"],3), "K1 in-loop ms", round(r["avg_launch_ms"],3), "alone", round(r["alone"]["avg_launch_ms"],3), j["calibration"])'
echo "default: $($B | tee $out/default.json | python3 -c "$pick")" >> $out/sweep.txt
echo "no-overlap: $($B --no-overlap | python3 -c "$pick")" >> $out/sweep.txt
for tile in 22 21; do for cap in 0 1 2 3 4; do
  echo "TT_TILE=$tile cap=$cap: $(TT_TILE=$tile $B --k1-blocks-per-cu $cap | python3 -c "$pick")" >> $out/sweep.txt
done; done
cat $out/sweep.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o ks -- python3 $root/bench.py --config c4 --no-cpu-baseline --steps 12 --warmup 4 > /dev/null 2>&1 || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/c4_default_kernel_stats.csv
python3 $root/tools/timeline.py $(find $out/trace -name "*kernel_trace.csv" | head -1) 8 2 > $out/c4_timeline.txt
rm -rf $out/trace
P="python3 $root/bench.py --config c4 --no-cpu-baseline --no-calibrate --steps 4 --warmup 2"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $out/pmc_tcc -o p -- $P > /dev/null 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -o p -- $P > /dev/null 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- $P > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- $P > /dev/null 2>&1 || exit 1
python3 - $out <<'PY' > $out/c4_pmc_per_kernel.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_tcc", "pmc_sq", "pmc_fetch", "pmc_write"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            key = ("K1 per-text" if "pool_norm" in n else "gemm NT relu" if "gemm_f32_kernel<0, 2" in n else "gemm NT" if "gemm_f32_kernel<0" in n else
                   "gemm NN" if "gemm_f32_kernel<1" in n else "gemm TN" if "gemm_f32_kernel<2" in n else "dW2+dH" if "gemm_tn_nn" in n else
                   "triplet" if "triplet_kernel" in n else "adam" if "adam_kernel" in n else None)
            if key:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for n, v in c.items()}
    hit = m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print(f"{k:13s} L2 hit {hit:5.3f}  fetch {m.get('FETCH_SIZE', 0) * 2048 / 1e6:9.1f} MB (x2)  write {m.get('WRITE_SIZE', 0) * 1024 / 1e6:9.1f} MB  "
          f"mfma_busy/SIMD/kernel_cycles {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(gui, 1):5.3f}  kernel_cycles {gui:10.0f}  "
          f"waves {m.get('SQ_WAVES', 0):8.0f} resident waves/CU {m.get('SQ_WAVE_CYCLES', 0) * 4 / max(gui, 1) / 256:5.1f}  "
          f"wait_any/wave_cycles {m.get('SQ_WAIT_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1):5.3f}")
PY
rm -rf $out/pmc_tcc $out/pmc_sq $out/pmc_fetch $out/pmc_write
cat $out/c4_pmc_per_kernel.txt
