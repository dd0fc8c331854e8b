import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd import _lib
lib = pkg.load()
for M, D, P in ((3072, 384, 512), (6144, 384, 384)):
    g = torch.Generator().manual_seed(1)
    dOut = (torch.randn(M, P, generator=g) / M).cuda(); X = torch.randn(M, D, generator=g).cuda()
    H = torch.relu(torch.randn(M, P, generator=g)).cuda(); W2 = torch.randn(P, P, generator=g).cuda()
    outs = []
    for bal in (0, 1):
        n = int(lib.tt_proj_bwd_workspace_bytes(M, D, P))
        ws = torch.zeros(n // 4, device="cuda")
        dW1 = torch.empty(P * D + P + P * P + P, device="cuda")
        t = _lib.make_tuning({"proj_balance": bal}) if bal else None
        rc = lib.tt_proj_bwd_f32(dOut.data_ptr(), X.data_ptr(), H.data_ptr(), W2.data_ptr(), dW1.data_ptr(), dW1[P * D:].data_ptr(),
                                 dW1[P * D + P:].data_ptr(), dW1[P * D + P + P * P:].data_ptr(), ws.data_ptr(), n, 0, M, D, P,
                                 C.byref(t) if t is not None else None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append((ws[: M * P].view(M, P).clone(), dW1.clone(), rc))
    a, b = outs[0][0], outs[1][0]
    ref = (dOut.double() @ W2.double()) * (H > 0)
    diff = (a - b).abs()
    bad = (diff > 1e-7 * max(1.0, float(a.abs().max()))).nonzero()
    print(M, D, P, "rc", outs[0][2], outs[1][2], "max |dH default - ref|", float((a.double() - ref).abs().max()), "max |dH bal - ref|", float((b.double() - ref).abs().max()),
          "n differing", int((diff > 0).sum()), "n bad", bad.shape[0])
    if bad.shape[0]:
        rows = bad[:, 0].unique(); cols = bad[:, 1].unique()
        print("  bad rows", rows[:8].tolist(), "...", rows[-4:].tolist(), "count", rows.numel(), " bad cols", cols[:8].tolist(), "...", cols[-4:].tolist(), "count", cols.numel())
        r0 = int(rows[0]); print("  sample", a[r0, cols[:4]].tolist(), b[r0, cols[:4]].tolist(), ref[r0, cols[:4]].tolist())
