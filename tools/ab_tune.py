#!/usr/bin/env python3
"""Interleaved A/B of bench.py variants on ONE box (boxes of the pool differ by +-4 %, so only same-box, interleaved
runs decide anything).  Every variant is a fresh `python bench.py` process; variants differ in their bench arguments —
typically `--tune FIELD=VALUE` (tt_tuning: per-engine, nothing process-wide).

    tools/ab_tune.py [--rounds 3] [--out gpurun_out/ab.txt] -v "base:" -v "t22:--tune proj_tile=22" -- --config c3 --steps 200

Prints one line per run (step time, K1 in the loop, streaming-K1 census from bench.py --debug-trace) and the per-variant medians."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    common = []
    if "--" in argv:
        i = argv.index("--")
        argv, common = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--out", default=None)
    ap.add_argument("-v", "--variant", action="append", required=True, help="label:extra bench args")
    a = ap.parse_args(argv)
    variants = [(v.split(":", 1)[0], v.split(":", 1)[1].split()) for v in a.variant]
    res = {k: [] for k, _ in variants}
    lines = []
    env = dict(os.environ)
    for rnd in range(a.rounds):
        for label, extra in variants:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--debug-trace", "10"] + common + extra,
                               capture_output=True, text=True, env=env, cwd=ROOT)
            if r.returncode != 0:
                print(f"{label}: bench failed\n{r.stderr[-2000:]}", file=sys.stderr)
                sys.exit(1)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            rf = j.get("roofline") or {}
            census = [l.split("planned):", 1)[1].strip() for l in r.stderr.splitlines() if "census" in l]
            rec = (j["ms_per_step"] * 1e3, rf.get("avg_launch_ms", 0.0) * 1e3, (rf.get("alone") or {}).get("avg_launch_ms", 0.0) * 1e3)
            res[label].append(rec)
            line = f"round {rnd} {label:14s} step {rec[0]:9.1f} us  K1 in loop {rec[1]:8.1f} us  alone {rec[2]:8.1f} us  {census[0] if census else ''}"
            print(line, flush=True)
            lines.append(line)
    lines.append("# medians: " + " ".join(common))
    for label, _ in variants:
        m = [statistics.median(x[i] for x in res[label]) for i in range(3)]
        lines.append(f"{label:14s} step {m[0]:9.1f} us  K1 in loop {m[1]:8.1f} us  alone {m[2]:8.1f} us   (n = {len(res[label])}, "
                     f"step min {min(x[0] for x in res[label]):.1f} max {max(x[0] for x in res[label]):.1f})")
    print("\n".join(lines[-len(variants) - 1:]))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "a") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
