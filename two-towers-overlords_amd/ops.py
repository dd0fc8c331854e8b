"""Tensor-level entry points over the C ABI (include/tt_hotpath.h).

Shapes, dtypes, contiguity and devices are validated here, *before* the call;
the C side only sees raw device pointers and sizes.  All work is enqueued on
torch's current HIP stream.  There is no fallback implementation: without a
GPU tensor and the HIP library these functions raise.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import TT_I32, TT_I64, call

_INT_WIDTH = {torch.int32: TT_I32, torch.int64: TT_I64}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_SWITCH = _NoSwitch()


def _on(device: torch.device):
    """Context that makes `device` current for the launch — a no-op object when it already is (entering
    `torch.cuda.device` costs several microseconds of host time per call, and the module path makes ~20 calls per step)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(device)


def _req(t: torch.Tensor, name: str, dtype=None, ndim: Optional[int] = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name}: expected a GPU (HIP) tensor, got device {t.device}. The MI355X HIP library is the "
            "only implementation of this path; there is no CPU fallback."
        )
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous (shape {tuple(t.shape)}, stride {t.stride()})")
    return t


def _int_tensor(t: torch.Tensor, name: str) -> int:
    if t.dtype not in _INT_WIDTH:
        raise TypeError(f"{name}: expected int32 or int64, got {t.dtype}")
    return _INT_WIDTH[t.dtype]


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _mask_or_lens(mask, lens, N: int, S: int, dev) -> Tuple[Optional[torch.Tensor], int, Optional[torch.Tensor]]:
    mw = TT_I64
    if mask is not None:
        _req(mask, "attention_mask", ndim=2)
        mw = _int_tensor(mask, "attention_mask")
        if tuple(mask.shape) != (N, S) or mask.device != dev:
            raise ValueError(f"attention_mask: expected shape {(N, S)} on {dev}, got {tuple(mask.shape)} on {mask.device}")
        lens = None
    elif lens is not None:
        _req(lens, "lens", dtype=torch.int32, ndim=1)
        if lens.shape[0] != N or lens.device != dev:
            raise ValueError(f"lens: expected shape {(N,)} on {dev}")
    return mask, mw, lens


_SCRATCH: dict = {}


def _scratch(device: torch.device, nbytes: int) -> torch.Tensor:
    """Grow-only per-device scratch for kernels whose workspace lives only inside one call
    (stream-ordered reuse: every user launches on the current stream)."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        _SCRATCH[key] = buf
    return buf


def gather_pool_norm(
    table: torch.Tensor,
    input_ids: torch.Tensor,
    attention_mask: Optional[torch.Tensor] = None,
    lens: Optional[torch.Tensor] = None,
    out: Optional[torch.Tensor] = None,
    check_ids: bool = False,
    sliced: Optional[bool] = None,
    stream_form: Optional[int] = None,
    form: Optional[str] = None,
    tuning=None,
) -> torch.Tensor:
    """K1: normalize(mean_pool(table[input_ids], attention_mask)) -> [N, D].
    Replaces backend/model.py:51-56 (`embed` mode).  check_ids=True synchronises
    and raises if any unmasked id was outside [0, V).  sliced: force the per-text (False) or the
    XCD-sliced (True, tt_gather_pool_norm_sliced_f32) form; None picks the faster one for the
    shape (tt_k1_prefers_sliced).  form: "rows" / "columns" picks the slicing of the sliced form (default: the
    measured rule, tt_k1_sliced_form; either implies sliced=True), "per_text" = sliced=False, "binned" = tokens sorted
    by row class + column slices (tt_gather_pool_norm_binned_f32: big tables named many times per launch).  Same result to fp32
    summation order.  stream_form=n: the streaming form (tt_gather_pool_norm_stream_f32: persistent workgroups, at
    most n per compute unit; bitwise the per-text result; `tuning`: k1_stream_rows / _offer) — what the
    engine uses for tables that stream from HBM."""
    if form is not None:
        if form not in ("per_text", "rows", "columns", "binned"):
            raise ValueError("form: 'per_text', 'rows', 'columns' or 'binned'")
        sliced = form != "per_text"
    _req(table, "table", torch.float32, 2)
    _req(input_ids, "input_ids", ndim=2)
    iw = _int_tensor(input_ids, "input_ids")
    if input_ids.device != table.device:
        raise ValueError("input_ids and table must be on the same device")
    N, S = input_ids.shape
    V, D = table.shape
    mask, mw, lens = _mask_or_lens(attention_mask, lens, N, S, table.device)
    if out is None:
        out = torch.empty((N, D), dtype=torch.float32, device=table.device)
    else:
        _req(out, "out", torch.float32, 2)
        if tuple(out.shape) != (N, D):
            raise ValueError(f"out: expected {(N, D)}, got {tuple(out.shape)}")
    flag = torch.zeros(1, dtype=torch.int32, device=table.device) if check_ids else None
    with _on(table.device):
        if form == "binned":  # tokens sorted by row class + 32-column slices (tt_gather_pool_norm_binned_f32)
            nbytes = int(_lib.load().tt_gather_pool_binned_workspace_bytes(N, D, S))
            if nbytes == 0:
                raise ValueError("form='binned' needs D % 32 == 0, D <= 1024, S <= 512")
            ws = _scratch(table.device, nbytes)
            call("tt_gather_pool_norm_binned_f32", table.data_ptr(), input_ids.data_ptr(), iw, _ptr(mask), mw, _ptr(lens),
                 out.data_ptr(), N, S, D, V, _ptr(flag), ws.data_ptr(), nbytes, _stream())
        elif stream_form:
            nbytes = int(_lib.load().tt_gather_pool_stream_workspace_bytes())
            ws = _scratch(table.device, nbytes)  # queue head + per-CU registry; stream-ordered reuse
            call("tt_gather_pool_norm_stream_f32", table.data_ptr(), input_ids.data_ptr(), iw, _ptr(mask), mw,
                 _ptr(lens), out.data_ptr(), N, S, D, V, _ptr(flag), ws.data_ptr(), nbytes, int(stream_form),
                 _lib.tuning_ref(tuning), _stream())
        else:
            if sliced is None:
                sliced = bool(_lib.load().tt_k1_prefers_sliced(V, D, N, S)) and table.data_ptr() % 16 == 0
            if sliced:
                nbytes = int(_lib.load().tt_gather_pool_sliced_workspace_bytes(N, D))
                ws = _scratch(table.device, nbytes)
                call("tt_gather_pool_norm_sliced_f32", table.data_ptr(), input_ids.data_ptr(), iw, _ptr(mask), mw,
                     _ptr(lens), out.data_ptr(), N, S, D, V, _ptr(flag), ws.data_ptr(), nbytes,
                     _lib.K1_FORMS[form] if form in ("rows", "columns") else _lib.TT_K1_AUTO, _stream())
            else:
                call("tt_gather_pool_norm_f32", table.data_ptr(), input_ids.data_ptr(), iw, _ptr(mask), mw, _ptr(lens),
                     out.data_ptr(), N, S, D, V, _ptr(flag), _stream())
    if check_ids and int(flag.item()) != 0:
        raise IndexError(f"input_ids contains a token id outside [0, {V})")
    return out


def pool_norm_dense(
    hidden: torch.Tensor,
    attention_mask: Optional[torch.Tensor] = None,
    lens: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """normalize(mean_pool(hidden, attention_mask)) over dense [N,S,D] states
    (`bert` mode).  Replaces backend/model.py:55-56,63-72."""
    _req(hidden, "hidden", torch.float32, 3)
    N, S, D = hidden.shape
    mask, mw, lens = _mask_or_lens(attention_mask, lens, N, S, hidden.device)
    out = torch.empty((N, D), dtype=torch.float32, device=hidden.device)
    with _on(hidden.device):
        call("tt_pool_norm_dense_f32", hidden.data_ptr(), _ptr(mask), mw, _ptr(lens), out.data_ptr(), N, S, D, _stream())
    return out


def embed_gather(table: torch.Tensor, input_ids: torch.Tensor, check_ids: bool = False) -> torch.Tensor:
    """table[input_ids] -> [..., D] (materialised; backend/model.py:51-52 `embed` mode)."""
    _req(table, "table", torch.float32, 2)
    _req(input_ids, "input_ids")
    iw = _int_tensor(input_ids, "input_ids")
    V, D = table.shape
    out = torch.empty((*input_ids.shape, D), dtype=torch.float32, device=table.device)
    flag = torch.zeros(1, dtype=torch.int32, device=table.device) if check_ids else None
    with _on(table.device):
        call("tt_embed_gather_f32", table.data_ptr(), input_ids.data_ptr(), iw, out.data_ptr(), input_ids.numel(), D, V,
             _ptr(flag), _stream())
    if check_ids and int(flag.item()) != 0:
        raise IndexError(f"input_ids contains a token id outside [0, {V})")
    return out


def proj_fwd(x, w1, b1, w2, b2, tuning=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """K2 forward: returns (out, h) with h = relu(x w1^T + b1), out = h w2^T + b2.
    Replaces backend/model.py:33-38,59.  `tuning` (dict / TTTuning): proj_tile, proj_bk, xcd_order."""
    _req(x, "x", torch.float32, 2)
    M, D = x.shape
    P = w1.shape[0]
    for t, name, shape in ((w1, "w1", (P, D)), (b1, "b1", (P,)), (w2, "w2", (P, P)), (b2, "b2", (P,))):
        _req(t, name, torch.float32)
        if tuple(t.shape) != shape:
            raise ValueError(f"{name}: expected {shape}, got {tuple(t.shape)}")
    h = torch.empty((M, P), dtype=torch.float32, device=x.device)
    out = torch.empty((M, P), dtype=torch.float32, device=x.device)
    with _on(x.device):
        call("tt_proj_fwd_f32", x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
             h.data_ptr(), out.data_ptr(), M, D, P, _lib.tuning_ref(tuning), _stream())
    return out, h


def proj_bwd(d_out, x, h, w2, tuning=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """K2 backward for the parameters: returns (dW1, db1, dW2, db2).  `tuning`: proj_tile, proj_bk, tn_tile, xcd_order."""
    _req(d_out, "d_out", torch.float32, 2)
    _req(x, "x", torch.float32, 2)
    _req(h, "h", torch.float32, 2)
    _req(w2, "w2", torch.float32, 2)
    M, P = d_out.shape
    D = x.shape[1]
    if tuple(x.shape) != (M, D) or tuple(h.shape) != (M, P) or tuple(w2.shape) != (P, P):
        raise ValueError("proj_bwd: inconsistent shapes")
    dev = x.device
    # one buffer in the tower's flat layout (dW1 | db1 | dW2 | db2): the slab sums then leave in ONE launch
    flat = torch.empty(P * D + P + P * P + P, dtype=torch.float32, device=dev)
    dW1 = flat[: P * D].view(P, D)
    db1 = flat[P * D: P * D + P]
    dW2 = flat[P * D + P: P * D + P + P * P].view(P, P)
    db2 = flat[P * D + P + P * P:]
    if M == 0:
        for t in (dW1, db1, dW2, db2):
            t.zero_()
        return dW1, db1, dW2, db2
    nbytes = _lib.load().tt_proj_bwd_workspace_bytes(M, D, P)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
    with _on(dev):
        call("tt_proj_bwd_f32", d_out.data_ptr(), x.data_ptr(), h.data_ptr(), w2.data_ptr(), dW1.data_ptr(),
             db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(), ws.data_ptr(), nbytes, 0, M, D, P, _lib.tuning_ref(tuning),
             _stream())
    return dW1, db1, dW2, db2


def triplet_fwd_bwd(q, p, n, margin: float, grad_scale: Optional[float] = None, need_grad: bool = True):
    """K3: returns (loss[1], pos_dist[B], neg_dist[B], dq, dp, dn); grads are
    grad_scale * d(sum hinge) (default 1/B == loss.backward()).  Replaces
    backend/model.py:132-145."""
    for t, name in ((q, "anchor"), (p, "positive"), (n, "negative")):
        _req(t, name, torch.float32, 2)
    B, P = q.shape
    if tuple(p.shape) != (B, P) or tuple(n.shape) != (B, P):
        raise ValueError("triplet: anchor/positive/negative shapes differ")
    if B == 0:
        raise ValueError("triplet: empty batch")
    dev = q.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    pd = torch.empty(B, dtype=torch.float32, device=dev)
    nd = torch.empty(B, dtype=torch.float32, device=dev)
    hinge = torch.empty(B, dtype=torch.float32, device=dev)
    dq = dp = dn = None
    if need_grad:
        dq, dp, dn = torch.empty((3, B, P), dtype=torch.float32, device=dev).unbind(0)  # one buffer: see _TripletLossFn
    gs = (1.0 / B) if grad_scale is None else grad_scale
    with _on(dev):
        call("tt_triplet_fwd_bwd_f32", q.data_ptr(), p.data_ptr(), n.data_ptr(), margin, gs, B, P, loss.data_ptr(),
             pd.data_ptr(), nd.data_ptr(), hinge.data_ptr(), _ptr(dq), _ptr(dp), _ptr(dn), _stream())
    return loss, pd, nd, dq, dp, dn


def adam_step_flat(params, grads, exp_avg, exp_avg_sq, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8,
                   grad_scale: float = 1.0) -> None:
    """K4: one torch.optim.Adam step (defaults of backend/training.py:436) over flat fp32 buffers, in place."""
    n = params.numel()
    for t, name in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t, name, torch.float32)
        if t.numel() != n:
            raise ValueError(f"{name}: expected {n} elements, got {t.numel()}")
    with _on(params.device):
        call("tt_adam_f32", params.data_ptr(), grads.data_ptr(), 1, 0, exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n,
             lr, beta1, beta2, eps, step, grad_scale, _stream())


def adam_step_multi(tensors, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8) -> None:
    """K4 over separate tensors in ONE launch (tt_adam_multi_f32): `tensors` = [(param, grad, exp_avg, exp_avg_sq), ...],
    each a contiguous fp32 GPU tensor of the same numel; all take optimizer step `step` (1-based)."""
    if not tensors:
        return
    arr = (_lib.TTAdamTensor * len(tensors))()
    dev = tensors[0][0].device
    for i, quad in enumerate(tensors):
        n = quad[0].numel()
        for t, name in zip(quad, ("param", "grad", "exp_avg", "exp_avg_sq")):
            _req(t, name, torch.float32)
            if t.numel() != n or t.device != dev:
                raise ValueError(f"{name} of tensor {i}: expected {n} elements on {dev}")
        arr[i].param, arr[i].grad, arr[i].exp_avg, arr[i].exp_avg_sq = (t.data_ptr() for t in quad)
        arr[i].n = n
    with _on(dev):
        call("tt_adam_multi_f32", arr, len(tensors), lr, beta1, beta2, eps, int(step), _stream())


def cosine_distance(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """1 - cosine_similarity(x, y, dim=1) -> [B] (backend/model.py:132-135), forward only."""
    _req(x, "x", torch.float32, 2)
    _req(y, "y", torch.float32, 2)
    if x.shape != y.shape or x.device != y.device:
        raise ValueError("cosine_distance: x and y must have the same [B, P] shape and device")
    B, P = x.shape
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    with _on(x.device):
        call("tt_cosine_distance_f32", x.data_ptr(), y.data_ptr(), B, P, out.data_ptr(), _stream())
    return out


def cosine_distance_bwd(x: torch.Tensor, y: torch.Tensor, grad_dist: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dx, dy) = grad_dist[b] * d(1 - cos(x_b, y_b)) / d{x_b, y_b}."""
    _req(x, "x", torch.float32, 2)
    _req(y, "y", torch.float32, 2)
    _req(grad_dist, "grad_dist", torch.float32, 1)
    B, P = x.shape
    if tuple(y.shape) != (B, P) or grad_dist.shape[0] != B:
        raise ValueError("cosine_distance_bwd: inconsistent shapes")
    dx, dy = torch.empty_like(x), torch.empty_like(y)
    with _on(x.device):
        call("tt_cosine_distance_bwd_f32", x.data_ptr(), y.data_ptr(), grad_dist.data_ptr(), B, P, dx.data_ptr(), dy.data_ptr(),
             _stream())
    return dx, dy


def cosine_scores(q: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """out[i, j] = cosine_similarity(q[i], d[j]) for all pairs -> [Q, N] (MFMA GEMM on unit rows).
    Replaces the per-query `torch.cosine_similarity(query_embed.cpu(), doc_embeds.cpu(), dim=1)`
    of backend/training.py:297-299."""
    _req(q, "q", torch.float32, 2)
    _req(d, "d", torch.float32, 2)
    Q, P = q.shape
    N = d.shape[0]
    if d.shape[1] != P or d.device != q.device:
        raise ValueError("cosine_scores: q [Q,P] and d [N,P] must share P and the device")
    out = torch.empty((Q, N), dtype=torch.float32, device=q.device)
    nbytes = _lib.load().tt_cosine_scores_workspace_bytes(Q, N, P)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=q.device)
    with _on(q.device):
        call("tt_cosine_scores_f32", q.data_ptr(), d.data_ptr(), out.data_ptr(), Q, N, P, ws.data_ptr(), nbytes, _stream())
    return out


def ndcg_at_k(scores: torch.Tensor, relevance: torch.Tensor, ks=(10,), valid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Q, len(ks)] NDCG@k per query row with sklearn.metrics.ndcg_score semantics (tie-averaged
    DCG, ideal DCG from the sorted gains) — backend/training.py:301-311.  `valid` (bool/uint8 [Q,N])
    restricts each row to its candidate pool."""
    import ctypes as C

    _req(scores, "scores", torch.float32, 2)
    Q, N = scores.shape
    gains = relevance.to(torch.float32).contiguous()
    if tuple(gains.shape) != (Q, N) or gains.device != scores.device:
        raise ValueError("ndcg_at_k: relevance must be [Q, N] on the scores' device")
    vm = None
    if valid is not None:
        vm = valid.to(torch.uint8).contiguous()
        if tuple(vm.shape) != (Q, N) or vm.device != scores.device:
            raise ValueError("ndcg_at_k: valid must be [Q, N] on the scores' device")
    ks = [int(k) for k in ks]
    if not 1 <= len(ks) <= 8 or min(ks) < 1:
        raise ValueError("ndcg_at_k: 1..8 cut-offs, each >= 1")
    out = torch.empty((Q, len(ks)), dtype=torch.float32, device=scores.device)
    ks_dev = torch.empty(len(ks), dtype=torch.int32, device=scores.device)
    ks_host = (C.c_int32 * len(ks))(*ks)
    with _on(scores.device):
        call("tt_ndcg_at_k_f32", scores.data_ptr(), gains.data_ptr(), _ptr(vm), Q, N, ks_host, len(ks), ks_dev.data_ptr(),
             out.data_ptr(), _stream())
        torch.cuda.current_stream().synchronize()  # ks_host must outlive the async H2D copy
    return out


def inbatch_triplet_fwd_bwd(q: torch.Tensor, d: torch.Tensor, margin: float, pos_index: Optional[torch.Tensor] = None,
                            qid_q: Optional[torch.Tensor] = None, qid_d: Optional[torch.Tensor] = None,
                            grad_scale: Optional[float] = None, loss_denominator: Optional[float] = None,
                            dq_out: Optional[torch.Tensor] = None, dd_out: Optional[torch.Tensor] = None,
                            workspace: Optional[torch.Tensor] = None):
    """In-batch-negatives triplet loss: returns (loss[1], dq [Bq,P], dd [Bd,P]).
    Defaults: mean over the Bq*(Bd-1) (query, other document) pairs.  `workspace`: a float32 buffer of at least
    tt_inbatch_workspace_bytes(Bq, Bd, P) bytes the caller keeps between steps (the similarity matrix alone is
    Bq x Bd floats: taking ~100 MB from the caching allocator every step makes the host time of a step bimodal)."""
    _req(q, "q", torch.float32, 2)
    _req(d, "d", torch.float32, 2)
    Bq, P = q.shape
    Bd = d.shape[0]
    if d.shape[1] != P or d.device != q.device:
        raise ValueError("inbatch: q [Bq,P] and d [Bd,P] must share P and the device")
    if pos_index is None and Bq != Bd:
        raise ValueError("inbatch: pos_index is required when Bq != Bd")
    if pos_index is not None:
        _req(pos_index, "pos_index", torch.int32, 1)
    if (qid_q is None) != (qid_d is None):
        raise ValueError("inbatch: give both qid_q and qid_d or neither")
    if qid_q is not None:
        _req(qid_q, "qid_q", torch.int64, 1)
        _req(qid_d, "qid_d", torch.int64, 1)
    if Bq < 1 or Bd < 2:
        raise ValueError("inbatch: needs at least one query and two documents (every negative is another document)")
    z = float(Bq * (Bd - 1))
    gs = (1.0 / z) if grad_scale is None else grad_scale
    den = z if loss_denominator is None else loss_denominator
    loss = torch.empty(1, dtype=torch.float32, device=q.device)
    dq = torch.empty_like(q) if dq_out is None else _req(dq_out, "dq_out", torch.float32, 2)
    dd = torch.empty_like(d) if dd_out is None else _req(dd_out, "dd_out", torch.float32, 2)
    if tuple(dq.shape) != (Bq, P) or tuple(dd.shape) != (Bd, P):
        raise ValueError("inbatch: dq_out / dd_out must match q / d")
    nbytes = _lib.load().tt_inbatch_workspace_bytes(Bq, Bd, P)
    if workspace is not None:
        ws = _req(workspace, "workspace", torch.float32)
        if ws.numel() * 4 < nbytes or ws.device != q.device:
            raise ValueError(f"inbatch: workspace holds {ws.numel() * 4} bytes, {nbytes} are needed")
    else:
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=q.device)
    with _on(q.device):
        call("tt_inbatch_triplet_fwd_bwd_f32", q.data_ptr(), d.data_ptr(), _ptr(pos_index), _ptr(qid_q), _ptr(qid_d), margin,
             gs, den, Bq, Bd, P, loss.data_ptr(), dq.data_ptr(), dd.data_ptr(), 0, ws.data_ptr(), nbytes, _stream())
    return loss, dq, dd
