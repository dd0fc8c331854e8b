#!/bin/bash
# A/B of three builds (ab_libs/*.so): base, the unweighted column-sliced K1 at 57 registers, the same padded to 66
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s15; mkdir -p $out
cd $root
python -m pytest tests/test_gpu_parity.py -x -q -k "pool or gather or k1 or pad_id or K1 or sliced" > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log; tail -2 $out/tests.log
show() { python -c "
import json,sys; j=json.loads(open('$1').read().strip().splitlines()[-1]); r=j['roofline']
print('$2', round(j['ms_per_step']*1e3,1), 'K1 in loop', round(r['avg_launch_ms']*1e3,1), 'alone', round(r['alone']['avg_launch_ms']*1e3,1))"; }
for i in 1 2 3; do
  for v in base unw57 unw72; do
    python bench.py --no-cpu-baseline --no-secondary --steps 200 --lib ab_libs/$v.so > $out/c2_${v}_$i.json 2>/dev/null && show $out/c2_${v}_$i.json "c2 $v"
  done
done
for i in 1 2; do
  for v in base unw57 unw72; do
    python bench.py --no-cpu-baseline --no-secondary --config c3 --steps 200 --lib ab_libs/$v.so > $out/c3_${v}_$i.json 2>/dev/null && show $out/c3_${v}_$i.json "c3 $v"
    python bench.py --no-cpu-baseline --no-secondary --ragged --steps 200 --lib ab_libs/$v.so > $out/c2r_${v}_$i.json 2>/dev/null && show $out/c2r_${v}_$i.json "c2-ragged $v"
  done
done
