# How the configs[1] step depends on whether a column slice of the table (V x 128 bytes) fits an XCD's 4 MiB L2 beside
# the projection chain's operands: the same step at other table heights (diagnostic, bench.py --vocab).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r04_vocab; mkdir -p $out; cd $root
for v in 4000 8000 15000 22000 30522; do
  python bench.py --no-cpu-baseline --no-secondary --vocab $v --tune k1_form=columns --steps 300 --warmup 30 2>/dev/null > $out/v$v.json
  python bench.py --no-cpu-baseline --no-secondary --vocab $v --tune k1_form=columns --inputs cache --steps 300 --warmup 30 2>/dev/null > $out/v${v}_cache.json
  python - $out/v$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"V {sys.argv[2]:>6}  slice {int(sys.argv[2]) * 128 / 1e6:5.2f} MB  step {d['ms_per_step'] * 1e3:7.1f} us  K1 in loop {r['avg_launch_ms'] * 1e3:7.1f}  alone {r['alone']['avg_launch_ms'] * 1e3:7.1f}")
PY
done
