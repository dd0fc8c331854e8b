#!/bin/bash
# Runs the bench modes reported in DESIGN.md section 5 on ONE box and writes the JSON lines, rocprofv3 kernel stats,
# timelines, MFMA utilisation and K1 fabric-traffic (PMC) summaries to gpurun_out/refresh/ (copied into profiles/ afterwards).
#   usage: tools/refresh_profiles.sh [rNN] [benches|profiles|all]   (every variant is a bench argument: nothing reads the
#   environment; the two halves fit one 20-minute gpurun call each)
r=${1:-r06}; part=${2:-all}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/refresh; mkdir -p $out
cd $root
B="python bench.py --no-cpu-baseline --no-secondary"
if [ "$part" != "profiles" ]; then
python bench.py > $out/${r}_bench_c2_default.json 2> $out/${r}_bench_c2_default.err || exit 1
$B --steps 20 --warmup 5 > $out/${r}_bench_c2_default_k20_w5.json
$B --no-overlap > $out/${r}_bench_c2_no_overlap.json
$B --ragged > $out/${r}_bench_c2_ragged.json
$B --zipf > $out/${r}_bench_c2_zipf.json
$B --inputs index > $out/${r}_bench_c2_inputs_index.json
$B --inputs cache > $out/${r}_bench_c2_inputs_cache.json
$B --e2e --steps 1000 > $out/${r}_bench_c2_e2e_epoch1000.json
$B --h2d > $out/${r}_bench_c2_h2d.json                                   # PCIe-inclusive: ids from pinned host memory every step
$B --path module --steps 400 --warmup 50 > $out/${r}_bench_c2_module_path.json                         # the reference's own torch.optim.Adam, routed
$B --path module --optimizer fused --steps 400 --warmup 50 > $out/${r}_bench_c2_module_path_fused_adam.json
$B --path module-unfused --steps 200 > $out/${r}_bench_c2_module_unfused.json
$B --path module --strings columns --steps 64 --epochs 12 > $out/${r}_bench_c2_strings_columns.json 2> /dev/null
$B --path module --strings lists --steps 64 --epochs 12 > $out/${r}_bench_c2_strings_lists.json 2> /dev/null
$B --path module --strings loader --steps 16 --epochs 6 > $out/${r}_bench_c2_strings_loader.json 2> /dev/null
$B --path module --strings lists --steps 4 --epochs 2 --text-cache 0 > $out/${r}_bench_c2_strings_lists_nocache.json 2> /dev/null
python bench.py --eval > $out/${r}_eval.json 2> /dev/null
python bench.py --run-training > $out/${r}_run_training.json 2> /dev/null   # the whole run_training call end to end, bert + embed
$B --loss inbatch > $out/${r}_bench_c2_inbatch_b2048.json
$B --loss inbatch --batch 4096 > $out/${r}_bench_inbatch_b4096.json
$B --loss inbatch --batch 512 > $out/${r}_bench_inbatch_b512_share.json
$B --config c3 > $out/${r}_bench_c3_share.json
$B --config c4 --steps 40 --warmup 8 > $out/${r}_bench_c4_full.json
$B --config c4 --steps 40 --warmup 8 --tune k1_form=stream > $out/${r}_bench_c4_stream_form.json
$B --config c4 --steps 40 --warmup 8 --zipf > $out/${r}_bench_c4_zipf.json      # real-text-like id skew: which K1 form the policy picks is in roofline.kernel
$B --config c4 --steps 40 --warmup 8 --ragged > $out/${r}_bench_c4_ragged.json  # ragged prefix lengths (queries U{8..S}, documents U{32..S})
$B --config c4 --steps 40 --warmup 8 --zipf --tune k1_form=stream > $out/${r}_bench_c4_zipf_stream_form.json
$B --config c4 --steps 40 --warmup 8 --ragged --tune k1_form=stream > $out/${r}_bench_c4_ragged_stream_form.json
$B --config c4 --steps 20 --warmup 4 --no-overlap > $out/${r}_bench_c4_no_overlap.json
$B --config c4 --steps 12 --warmup 4 --inputs index > $out/${r}_bench_c4_inputs_index.json
echo "benches done"
fi
if [ "$part" = "benches" ]; then exit 0; fi
cd /tmp && export TMPDIR=/tmp
prof() { tag=$1; first=$2; shift 2
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o ks -- python3 $root/bench.py --no-cpu-baseline --no-secondary "$@" > /dev/null 2>&1
  cp $(find $out/prof_$tag -name "*kernel_stats.csv" | head -1) $out/${r}_bench_${tag}_kernel_stats.csv
  python3 $root/tools/timeline.py $(find $out/prof_$tag -name "*kernel_trace.csv" | head -1) $first 2 > $out/${r}_timeline_${tag}.txt || true
  rm -rf $out/prof_$tag; }
prof c2_default 120
prof c3_share 120 --config c3
prof c4_full 12 --config c4 --steps 20 --warmup 4
prof c2_module_path 60 --path module --steps 100
echo "kernel traces done"
# MFMA utilisation of the projection kernels (one-stream runs: counter collection serialises kernels anyway)
rm -f $out/${r}_k2_mfma_util.txt
mfma() { tag=$1; label=$2; shift 2
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_$tag -o p -- python3 $root/bench.py --no-cpu-baseline --no-secondary --no-calibrate --no-overlap "$@" > /dev/null 2>&1 || return 1
  python3 - $out/pmc_$tag "$label" <<'PY' >> $out/${r}_k2_mfma_util.txt
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
    print(f"{k:56s} mfma_util {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(m.get('GRBM_GUI_ACTIVE', 0) / 8, 1):5.3f}   launches {len(c.get('GRBM_GUI_ACTIVE', []))}")
PY
  rm -rf $out/pmc_$tag; }
mfma c2 "configs[1] B=2048 D=P=384 (one stream)" --steps 40 --warmup 2
mfma c2fork "configs[1], fork backward (TN and NN as separate launches)" --steps 40 --warmup 2 --tune backward=fork
mfma c3 "configs[2] one GPU's share B=1024 P=512 (one stream)" --config c3 --steps 40 --warmup 2
mfma ib4096 "in-batch negatives B=4096 (configs[4] on one GPU; B x B products: NT with the hinge epilogue, NN, TN)" --loss inbatch --batch 4096 --steps 20 --warmup 2
mfma ib512 "in-batch negatives B=512 (one GPU's share of configs[4])" --loss inbatch --batch 512 --steps 40 --warmup 2
mfma c4bk16 "configs[3] with the tiles used beside the streaming gather (K-step 16, 128x128 TN, XCD order)" --config c4 --steps 6 --warmup 2 --tune proj_tile=22 --tune proj_bk=16 --tune tn_tile=22 --tune xcd_order=1
cat $out/${r}_k2_mfma_util.txt
# K1 fabric-side traffic: FETCH_SIZE / WRITE_SIZE in separate passes of bench.py --k1-only
traffic() { cfg=$1; alg=$2; kern=$3; tag=$4; shift 4
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$cfg -o f -- python3 $root/bench.py --config $cfg --k1-only --no-cpu-baseline --no-secondary --steps 8 --warmup 2 "$@" > /dev/null 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$cfg -o w -- python3 $root/bench.py --config $cfg --k1-only --no-cpu-baseline --no-secondary --steps 8 --warmup 2 "$@" > /dev/null 2>&1 || return 1
  python3 $root/tools/pmc_traffic.py $out/pmc_fetch_$cfg $out/pmc_write_$cfg $out/${r}_k1_traffic_$tag.json $alg $kern
  rm -rf $out/pmc_fetch_$cfg $out/pmc_write_$cfg; }
traffic c2 1220542464 pool_colsliced_kernel,pool_colnorm_kernel c2
traffic c4 38856032256 pool_binned_kernel,pool_bincombine_kernel,pool_bin_kernel c4
traffic c4 38856032256 pool_norm_stream_kernel c4_stream --tune k1_form=stream
# SURVEY section 8 f2 under the profiler: kernel stats + MFMA utilisation of the cosine-score GEMM
bash $root/tools/r06_eval_prof.sh $out/evalprof > /dev/null 2>&1 && cp $out/evalprof/r06_eval_kernel_stats.csv $out/${r}_eval_kernel_stats.csv && cp $out/evalprof/r06_eval_mfma_util.txt $out/${r}_eval_mfma_util.txt
# the data-parallel step under a stand-in exchange at configs[2]'s share (the table of DESIGN.md section 6)
cd $root && python tools/ab_tune.py --rounds 2 --out $out/${r}_dp_sim_c3.txt -v "single:" -v "x0:--simulate-allreduce-us 0" -v "x30:--simulate-allreduce-us 30" -v "x60:--simulate-allreduce-us 60" -v "x100:--simulate-allreduce-us 100" -- --config c3 --steps 200 --warmup 20 > /dev/null 2>&1
echo refreshed
