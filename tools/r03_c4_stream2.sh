#!/bin/bash
# Round 3, second pass at configs[3]: XCD-aware tile order + 128x128 TN tiles beside the streaming K1.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/c4stream2; mkdir -p $out
cd $root
python3 -m pytest tests/test_gpu_stream_policy.py -x -q -m gpu > $out/pytest_policy.txt 2>&1 || { tail -30 $out/pytest_policy.txt; exit 1; }
tail -2 $out/pytest_policy.txt
B="python3 bench.py --config c4 --no-cpu-baseline --steps 12 --warmup 4"
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["ms_per_step"],3), "K1 in-loop ms", round(r["avg_launch_ms"],3), "alone", round(r["alone"]["avg_launch_ms"],3), j["calibration"]["ms_per_step"])'
echo "default: $($B | python3 -c "$pick")" >> $out/sweep.txt
for xcd in 0 1; do for tn in 11 22; do for wgs in 1 2; do
  u=0; [ $wgs = 2 ] && u=4
  echo "xcd=$xcd tn=$tn wgs=$wgs U=$u: $(TT_XCD=$xcd TT_TN_TILE=$tn TT_K1_STREAM_WGS=$wgs TT_K1_STREAM_U=$u $B | python3 -c "$pick")" >> $out/sweep.txt
done; done; done
echo "no-overlap (chain alone + K1 alone): $($B --no-overlap | python3 -c "$pick")" >> $out/sweep.txt
cat $out/sweep.txt
cd /tmp && export TMPDIR=/tmp
P="python3 $root/bench.py --config c4 --no-cpu-baseline --no-calibrate --steps 4 --warmup 2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- $P > /dev/null 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -o p -- $P > /dev/null 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -o p -- $P > /dev/null 2>&1 || exit 1
python3 - $out <<'PY' > $out/c4_pmc_per_kernel.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_tcc", "pmc_sq", "pmc_fetch"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            key = ("K1 stream" if "pool_norm_stream" in n else "K1 per-text" if "pool_norm" in n else
                   ("gemm NT" if "<0," in n.split("_kernel")[-1][:5] else "gemm NN" if "<1," in n.split("_kernel")[-1][:5] else "gemm TN") if "gemm_f32" in n else
                   "triplet" if "triplet_kernel" in n else "adam" if "adam_kernel" in n else None)
            if key:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for n, v in c.items()}
    hit = m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print(f"{k:13s} L2 hit {hit:5.3f}  fetch {m.get('FETCH_SIZE', 0) * 2048 / 1e6:9.1f} MB (x2)  mfma_busy/SIMD/kernel_cycles {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(gui, 1):5.3f}  kernel_cycles {gui:10.0f}")
PY
rm -rf $out/pmc_tcc $out/pmc_sq $out/pmc_fetch
cat $out/c4_pmc_per_kernel.txt
