#!/usr/bin/env python3
"""Host-side profile of the drop-in route (bench.py --path module): cProfile over one timed epoch, top functions by
cumulative and by own time; plus the wall time of enqueueing alone (no device sync inside the epoch except the prints)."""
import cProfile, pstats, sys, os, io, contextlib
sys.argv = ["bench.py", "--path", "module", "--steps", "300", "--warmup", "20", "--no-cpu-baseline", "--no-secondary"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("tt_bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
pr = cProfile.Profile()
pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    b.main()
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[:48]))
