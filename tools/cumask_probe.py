#!/usr/bin/env python3
"""Probe: K1 and the projection chain on CU-masked streams (hipExtStreamCreateWithCUMask). Diagnostic."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd import ops
pkg.load()
hip = None
for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
    try:
        hip = ctypes.CDLL(name); break
    except OSError:
        pass
assert hip is not None
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int

def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask rc={rc}"
    return torch.cuda.ExternalStream(st.value)

dev = torch.device("cuda")
V, D, N, S = 30522, 384, 6144, 128
table = (torch.randn(V, D) * 0.02).to(dev)
g = torch.Generator().manual_seed(0)
ids = [torch.randint(0, V, (N, S), generator=g, dtype=torch.int32).to(dev) for _ in range(4)]
out = torch.empty(N, D, device=dev)
x = torch.randn(N, D, device=dev); w1 = torch.randn(D, D, device=dev) * .05; b1 = torch.zeros(D, device=dev)
w2 = torch.randn(D, D, device=dev) * .05; b2 = torch.zeros(D, device=dev)
masks = {
    "all 256": [0xffffffff] * 8,
    "alternate bits (128)": [0x55555555] * 8,
    "low 128 bits": [0xffffffff] * 4 + [0] * 4,
    "3 of 4 bits (192)": [0x77777777] * 8,
    "1 of 4 bits (64)": [0x11111111] * 8,
    "low 16 of each 32 (128)": [0x0000ffff] * 8,
}
for name, words in masks.items():
    st = masked_stream(words)
    with torch.cuda.stream(st):
        for kind in ("K1 sliced", "K1 per-text", "fwd pair"):
            def run(i):
                if kind == "K1 sliced": ops.gather_pool_norm(table, ids[i % 4], out=out, sliced=True)
                elif kind == "K1 per-text": ops.gather_pool_norm(table, ids[i % 4], out=out, sliced=False)
                else: ops.proj_fwd(x, w1, b1, w2, b2)
            for i in range(5): run(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for i in range(30): run(i)
            e1.record(st); st.synchronize()
            print(f"{name:28s} {kind:12s} {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us", flush=True)
