#!/bin/bash
# Round 6 stability record after the trainer split / engine stream recycling: engine lifecycles (incl. graph-replaying trainers
# between destroyed engines), a 20k-step soak of the default pipeline, bitwise determinism of the overlapped schedules, and the
# graph-crash scenario.   usage (GPU box): bash tools/r06_stability.sh > gpurun_out/r06_stability.txt 2>&1
set -o pipefail
echo "== tools/lifecycle.py (30 x two engines created / used / destroyed)"; timeout -k 10 300 python tools/lifecycle.py 2>&1 | grep -v amdgpu.ids || exit 1
echo "== tools/r06_graph_probe2.py engine (a routed epoch left as cyclic garbage, then eager / graph / graph trainers)"; timeout -k 10 300 python tools/r06_graph_probe2.py engine 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
echo "== tools/soak.py 40 (20,480 steps of the default pipeline)"; timeout -k 10 500 python tools/soak.py 40 2>&1 | grep -v amdgpu.ids || exit 1
echo "== tools/determinism_probe.py 300 triplet"; timeout -k 10 400 python tools/determinism_probe.py 300 triplet 2>&1 | grep -v amdgpu.ids | tail -8 || exit 1
echo "== tools/roles_determinism.py 200 (the drop-in route's three streams)"; timeout -k 10 300 python tools/roles_determinism.py 200 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
echo "stability ok"
