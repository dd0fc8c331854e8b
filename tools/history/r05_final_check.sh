#!/bin/bash
root=$GRAFT_REPO_ROOT; cd $root
t0=$(date +%s)
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_final_bench.json 2> gpurun_out/r05_final_bench.err
echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r05_final_bench.json").read().strip().splitlines()[-1])
r=j["roofline"]
print(j["metric"], "|", round(j["value"]/1e6,3), "M/s", round(j["ms_per_step"]*1e3,1), "us", "| bound", r["bound"], "frac", round(r["frac"],3), "nominal", round(r["nominal_hbm"]["frac"],3), "| traffic", r["traffic"], (r["traffic_source"] or "")[:32], "| k2", r["k2"] and r["k2"]["mfma_util_min"], "| cpu", round(j["cpu_baseline"]["value"]), j["cpu_baseline"]["cores"], "| secondary", round(j["secondary"]["ms_per_step"],3), round(j["secondary"]["roofline"]["frac"],3))
print(sorted(j.keys()))
print("dp_breakdown", {k: (round(v,4) if isinstance(v,float) else v) for k,v in j["dp_breakdown"].items() if k != "note"})
PY
