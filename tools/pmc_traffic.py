#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes for the K1 kernel into
profiles/<tag>.json (per-launch HBM-side traffic, corrected as MI355X_MICROARCH.md
§HBM prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half the bytes of a wide (16 B/lane) coalesced read -> x2; WRITE_SIZE is exact
for 16 B/lane stores).  The counters sit on the L2's fabric side: Infinity-Cache hits
are counted, so for a cache-resident table this is fabric traffic, not DRAM traffic."""
import csv
import glob
import json
import statistics
import sys


def per_launch(dirname, counter, kernel_substr):
    """Counter values of the launches whose kernel name contains `kernel_substr`; with several
    comma-separated substrings (a K1 made of two launches) the i-th launches of each are added."""
    names = kernel_substr.split(",")
    vals = {n: [] for n in names}
    for f in glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for n in names:
                if n in r["Kernel_Name"]:
                    vals[n].append(float(r["Counter_Value"]))
    k = min(len(v) for v in vals.values())
    return [sum(vals[n][i] for n in names) for i in range(k)]


def main():
    fetch_dir, write_dir, out, alg_bytes = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    kern = sys.argv[5] if len(sys.argv) > 5 else "pool_norm_kernel"
    fe = per_launch(fetch_dir, "FETCH_SIZE", kern)
    wr = per_launch(write_dir, "WRITE_SIZE", kern)
    fetch_b = statistics.mean(fe) * 1024 * 2
    write_b = statistics.mean(wr) * 1024
    d = {
        "kernel": kern, "launches": {"fetch_pass": len(fe), "write_pass": len(wr)},
        "FETCH_SIZE_KiB_mean": statistics.mean(fe), "WRITE_SIZE_KiB_mean": statistics.mean(wr),
        "fetch_bytes_corrected_x2": fetch_b, "write_bytes": write_b, "traffic_bytes_per_launch": fetch_b + write_b,
        "algorithmic_bytes_per_launch": alg_bytes, "traffic_over_algorithmic": (fetch_b + write_b) / alg_bytes,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --k1-only`; "
                  "bytes = KiB*1024, FETCH x2 (gfx950 128-B requests tallied at 64 B), MI355X_MICROARCH.md §HBM",
    }
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    main()
