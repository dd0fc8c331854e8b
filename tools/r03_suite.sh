#!/bin/bash
# whole GPU suite (one pytest process), the driver's smoke(), and the default bench line as the driver runs it
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/suite; mkdir -p $out
cd $root
python3 -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1; rc=$?
tail -6 $out/pytest_gpu.txt | cut -c1-300
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; rc2=$?
tail -2 $out/smoke.txt
( time python3 bench.py > $out/bench_default.json ) 2> $out/bench_default.time; rc3=$?
python3 -c 'import sys,json; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=j["secondary"]; print(j["value"], j["ms_per_step"], j["roofline"]["frac"], "sec:", s.get("value"), s.get("ms_per_step"), (s.get("roofline") or {}).get("frac"), "bert:", j["cpu_baseline_bert"]["value"])' $out/bench_default.json
tail -3 $out/bench_default.time
exit $((rc + rc2 + rc3))
