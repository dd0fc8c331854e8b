#!/usr/bin/env python3
"""cProfile of a bench.py run's host side (where the Python / ctypes / torch-API time of a step goes):
    python tools/hostprof_bench.py [bench args]      -> top 35 functions by cumulative time, on stderr"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
    print(s.getvalue(), file=sys.stderr)
