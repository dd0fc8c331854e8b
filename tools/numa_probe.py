#!/usr/bin/env python3
"""Where is the GPU relative to the CPUs this process may run on? (diagnostic)"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
p = torch.cuda.get_device_properties(0)
bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0" if hasattr(p, "pci_bus_id") else None
print("gpu", p.name, "bdf", bdf)
for path in glob.glob("/sys/bus/pci/devices/*/numa_node"):
    if bdf and bdf in path:
        print("gpu numa_node", open(path).read().strip())
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(n.split("/")[-2], open(n).read().strip())
def cur_cpu():
    try:
        return int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36])
    except Exception:
        return -1
print("affinity", len(os.sched_getaffinity(0)), "cpus; running on cpu", cur_cpu())
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd.training import TripletTrainer, TripletBatch
from two_towers_overlords_amd.engine import flat_param_slices
V, D, P, B, S = 30522, 384, 384, 2048, 128
dev = torch.device('cuda')
table = (torch.randn(V, D) * 0.02).to(dev)
params = {n: (torch.rand(*sh) * 0.1 - 0.05).to(dev) for n, _, sh in flat_param_slices(D, P)}
tr = TripletTrainer(table=table, params=params, max_batch=B, max_seq_len=S)
b = TripletBatch(ids=torch.randint(0, V, (3, B, S), dtype=torch.int32, device=dev))
def run(n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    issued = 0
    while issued < n and len(tr._pending) < 2:
        tr.gather(b); issued += 1
    while tr._pending:
        nxt = None
        if issued < n: nxt = b; issued += 1
        tr.advance(nxt)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n
run(50)
nodes = {}
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    cpus = set()
    for part in open(n).read().strip().split(","):
        a, _, bb = part.partition("-")
        cpus.update(range(int(a), int(bb or a) + 1))
    nodes[n.split("/")[-2]] = cpus
allowed = os.sched_getaffinity(0)
print("unpinned: host %.1f total %.1f us/step (cpu %d)" % (*run(), cur_cpu()))
for name, cpus in nodes.items():
    c = cpus & allowed
    if not c: continue
    os.sched_setaffinity(0, c)
    h, t = run()
    print(f"pinned to {name} ({len(c)} cpus): host {h:.1f} total {t:.1f} us/step (cpu {cur_cpu()})")
os.sched_setaffinity(0, allowed)
