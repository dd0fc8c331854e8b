#!/bin/bash
# round 4, first measurement pass after the tt_tuning refactor: new tests, then the bench lines that later work is
# measured against (one box, back to back).
set -o pipefail
O=gpurun_out/r04_first; mkdir -p $O
python -m pytest tests/test_gpu_model_api.py tests/test_gpu_parity.py -x -q -m gpu \
  -k "adam_multi or cosine_distance or pretokenised or accumulation or optimized or module_path" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
B="python bench.py --no-cpu-baseline --no-secondary"
$B --steps 200 --warmup 20 > $O/c2.json 2> $O/c2.err
$B --steps 200 --warmup 20 --tune tn_transpose=1 > $O/c2_tnt.json 2> $O/c2_tnt.err
$B --steps 200 --warmup 20 > $O/c2_b.json 2> $O/c2_b.err
$B --steps 200 --warmup 20 --tune tn_transpose=1 > $O/c2_tnt_b.json 2> $O/c2_tnt_b.err
$B --config c3 --steps 200 --warmup 20 > $O/c3.json 2> $O/c3.err
$B --config c3 --steps 200 --warmup 20 --tune tn_transpose=1 > $O/c3_tnt.json 2> $O/c3_tnt.err
$B --path module --steps 100 --warmup 20 > $O/c2_module.json 2> $O/c2_module.err
TT_TRACE=1 $B --config c4 --steps 30 --warmup 8 > $O/c4.json 2> $O/c4.err
TT_TRACE=1 $B --config c4 --steps 30 --warmup 8 --tune k1_stream_offer=2 > $O/c4_offer2.json 2> $O/c4_offer2.err
TT_TRACE=1 $B --config c4 --steps 30 --warmup 8 --tune k1_stream_offer=64 > $O/c4_offer64.json 2> $O/c4_offer64.err
TT_TRACE=1 $B --config c4 --steps 30 --warmup 8 --tune k1_stream_offer=16 --tune k1_stream_linger=8 > $O/c4_linger8.json 2> $O/c4_linger8.err
for f in c2 c2_tnt c2_b c2_tnt_b c3 c3_tnt c2_module c4 c4_offer2 c4_offer64 c4_linger8; do
  python - "$O/$f.json" <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); rf=r.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "%.2f M/s"%(r["value"]/1e6), "%.1f us"%(r["ms_per_step"]*1e3), "K1 in-loop %.3f ms frac %.3f alone %.3f"%(rf.get("avg_launch_ms",0), rf.get("frac",0), (rf.get("alone") or {}).get("frac",0)) if rf else "")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
grep -h "census" $O/c4*.err
