#!/bin/bash
# usage (on the GPU box): tools/clock_probe.sh TAG bench-args...  — samples rocm-smi clocks / power every ~100 ms while one
# bench run is in flight (diagnostic: are two kernels that share the chip sharing its POWER budget?)
tag=$1; shift
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/clock_probe; mkdir -p $out
( while true; do rocm-smi -d 0 --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done ) > $out/$tag.smi &
SMI=$!
python3 $root/bench.py --no-cpu-baseline --no-secondary "$@" > $out/$tag.json 2>/dev/null
kill $SMI
python3 - $out/$tag.smi $out/$tag.json $tag <<'PY'
import json, sys, re
rows = []
for l in open(sys.argv[1]):
    try:
        d = json.loads(l)["card0"]
    except Exception:
        continue
    sclk = next((v for k, v in d.items() if k.startswith("sclk")), "")
    pw = next((v for k, v in d.items() if "Power" in k and "W" in k), "")
    m = re.search(r"(\d+)\s*Mhz", sclk, re.I)
    try:
        rows.append((int(m.group(1)) if m else None, float(pw)))
    except Exception:
        rows.append((int(m.group(1)) if m else None, None))
busy = sorted(rows, key=lambda r: -(r[1] or 0))[: max(3, len(rows) // 3)]  # the third of the samples with the highest power
j = json.load(open(sys.argv[2]))
f = lambda xs: (sum(xs) / len(xs)) if xs else float("nan")
print(f"{sys.argv[3]:16s} step {j['ms_per_step'] * 1e3:8.1f} us   samples {len(rows):3d}   under load: sclk {f([r[0] for r in busy if r[0]]):6.0f} MHz  power {f([r[1] for r in busy if r[1]]):6.0f} W")
PY
