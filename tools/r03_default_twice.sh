#!/bin/bash
# The default bench line twice on one box: headline, `secondary` (configs[3]), both CPU baselines, wall time of the command.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/verify; mkdir -p $out
cd $root
for i in 1 2; do
t0=$(date +%s.%N)
python3 bench.py > $out/bench_default_$i.json 2> $out/bench_default_$i.err || { tail -5 $out/bench_default_$i.err; exit 1; }
t1=$(date +%s.%N)
python3 - $out/bench_default_$i.json $t0 $t1 <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]; s = j["secondary"]
print(round(j["value"] / 1e6, 3), "M/s", round(j["ms_per_step"] * 1e3, 1), "us; secondary", round(s["ms_per_step"], 3), "ms K1 frac",
      round(s["roofline"]["frac"], 3), "alone", round(s["roofline"]["alone"]["frac"], 3), "cpu", round(j["cpu_baseline"]["value"]),
      round(j["cpu_baseline_bert"]["value"]), "wall", round(float(sys.argv[3]) - float(sys.argv[2]), 1), "s")
PY
done
