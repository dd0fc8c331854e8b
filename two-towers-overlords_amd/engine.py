"""HotPathEngine — Python handle on tt_engine (include/tt_hotpath.h): the whole
batch body of backend/training.py:36-53 (three encodes, loss, backward, Adam) as
a few C calls over flat, torch-owned device buffers."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import TTEngineConfig, call
from .ops import _INT_WIDTH, _req


def tower_param_count(D: int, P: int) -> int:
    return P * D + P + P * P + P


def flat_param_slices(D: int, P: int):
    """(name, offset, shape) of the 8 trainable tensors inside the flat buffer —
    the reference's model.parameters() order (backend/model.py:33-38,84-85)."""
    out, off = [], 0
    for tower in ("query_tower", "document_tower"):
        for name, shape in (("projection.0.weight", (P, D)), ("projection.0.bias", (P,)),
                            ("projection.2.weight", (P, P)), ("projection.2.bias", (P,))):
            n = 1
            for s in shape:
                n *= s
            out.append((f"{tower}.{name}", off, shape))
            off += n
    return out


class HotPathEngine:
    def __init__(self, table: torch.Tensor, projection_dim: int, max_batch: int, max_seq_len: int,
                 k1_blocks_per_cu: int = 0, tuning=None, grad_slabs: int = 0, pack_roles: bool = False):
        _req(table, "table", torch.float32, 2)
        self.device = table.device if table.device.index is not None else torch.device("cuda", torch.cuda.current_device())
        self.table = table
        self.V, self.D = table.shape
        self.P = int(projection_dim)
        self.max_B, self.max_S = int(max_batch), int(max_seq_len)
        forced = _lib.make_tuning(tuning or {}).k1_form  # the partial-sum workspace is sized for a form a tuning will force
        cfg = TTEngineConfig(self.V, self.D, self.P, self.max_B, self.max_S, self.device.index, int(k1_blocks_per_cu),
                             int(grad_slabs), int(forced), 1 if pack_roles else 0)
        self.has_pack_roles = bool(pack_roles)
        lib = _lib.load()
        nbytes = lib.tt_engine_workspace_bytes(C.byref(cfg))
        if nbytes == 0:
            raise ValueError(f"unsupported engine configuration V={self.V} D={self.D} P={self.P} B={max_batch} S={max_seq_len}")
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            assert self.workspace.data_ptr() % 256 == 0
            h = C.c_void_p()
            call("tt_engine_create", C.byref(cfg), self.workspace.data_ptr(), nbytes, self._stream(), C.byref(h))
        self._h = h
        self.n_params = int(lib.tt_engine_param_count(h))
        self.n_slabs = int(lib.tt_engine_grad_slabs(h))
        self.params = self.exp_avg = self.exp_avg_sq = None
        self.tuning = None
        self.backward_mode = "auto"  # the schedule last set through set_backward (a tuning's `backward` field overrides it in the engine)
        if tuning:
            self.set_tuning(tuning)

    def set_tuning(self, tuning) -> None:
        """Bind a tt_tuning (dict of field -> value, or _lib.TTTuning; None / {} = all auto) to THIS engine: K1 form,
        streaming-gather residency, projection tiles, backward schedule ...  Speed only (include/tt_hotpath.h)."""
        t = _lib.make_tuning(tuning or {})
        call("tt_engine_set_tuning", self._h, C.byref(t))
        self.tuning = {k: getattr(t, k) for k in _lib.TUNING_FIELDS if getattr(t, k)}

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    _deferred = []  # (handle, workspace) of engines whose close() fell inside a stream capture: destroyed at the next safe close()

    def close(self) -> None:
        """Destroy the handle (synchronises the device).  Never inside a stream capture: a garbage-collected engine's
        `__del__` may run at any allocation — also between capture_begin and capture_end of somebody's hipGraph, where a
        device synchronise invalidates the capture (round 6: a graph captured that way crashed the process at its first
        replay).  There the handle and its workspace are parked and destroyed by the next close() outside a capture."""
        h = getattr(self, "_h", None)
        if not h:
            return
        self._h = None
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            HotPathEngine._deferred.append((h, self.workspace, self.device))
            return
        pending, HotPathEngine._deferred = HotPathEngine._deferred + [(h, self.workspace, self.device)], []
        for handle, _ws, dev in pending:
            torch.cuda.synchronize(dev)
            _lib.load().tt_engine_destroy(handle)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ binding
    def bind(self, params: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor) -> None:
        for t, name in ((params, "params"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
            _req(t, name, torch.float32, 1)
            if t.numel() != self.n_params or t.device != self.device:
                raise ValueError(f"{name}: expected {self.n_params} floats on {self.device}")
        self.params, self.exp_avg, self.exp_avg_sq = params, exp_avg, exp_avg_sq
        call("tt_engine_bind", self._h, self.table.data_ptr(), params.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr())

    # ------------------------------------------------------------------ phases
    def _ids_args(self, ids, mask, lens):
        _req(ids, "ids", ndim=3)
        if ids.dtype not in _INT_WIDTH or ids.shape[0] not in (2, 3) or ids.device != self.device:
            raise ValueError("ids: expected an int32/int64 [roles, B, S] tensor (q, pos[, neg]) on the engine device")
        R, B, S = ids.shape
        mw = 8
        if mask is not None:
            _req(mask, "mask", ndim=3)
            if mask.dtype not in _INT_WIDTH or tuple(mask.shape) != (R, B, S) or mask.device != self.device:
                raise ValueError("mask: expected an int32/int64 [roles, B, S] tensor")
            mw = _INT_WIDTH[mask.dtype]
            lens = None
        elif lens is not None:
            _req(lens, "lens", torch.int32, 2)
            if tuple(lens.shape) != (R, B) or lens.device != self.device:
                raise ValueError("lens: expected an int32 [roles, B] tensor")
        return (ids.data_ptr(), _INT_WIDTH[ids.dtype], None if mask is None else mask.data_ptr(), mw,
                None if lens is None else lens.data_ptr(), R, B, S)

    def pool(self, slot: int, ids, mask=None, lens=None, stream: Optional[int] = None) -> None:
        """K1 over ids [roles, B, S] (roles = 3: q, pos, neg; 2: q, pos with in-batch negatives by index).
        `stream`: raw hipStream_t to launch on (default: torch's current stream)."""
        a = self._ids_args(ids, mask, lens)
        call("tt_engine_pool", self._h, slot, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7],
             self._stream() if stream is None else stream)

    def _role_array(self, blocks):
        if len(blocks) not in (2, 3):
            raise ValueError("pool_roles: 2 or 3 roles (q, pos[, neg])")
        arr = (_lib.TTRoleTokens * len(blocks))()
        B = blocks[0][0].shape[0]
        for r, (ids, mask) in enumerate(blocks):
            _req(ids, "ids", ndim=2)
            if ids.dtype not in _INT_WIDTH or ids.device != self.device or ids.shape[0] != B:
                raise ValueError("pool_roles: ids must be int32/int64 [B, S] tensors on the engine device, one B for all roles")
            arr[r].ids, arr[r].ids_dtype, arr[r].S = ids.data_ptr(), _INT_WIDTH[ids.dtype], ids.shape[1]
            if mask is not None:
                _req(mask, "mask", ndim=2)
                if mask.dtype not in _INT_WIDTH or tuple(mask.shape) != tuple(ids.shape) or mask.device != self.device:
                    raise ValueError("pool_roles: mask must be an int32/int64 tensor of the ids' shape")
                arr[r].mask, arr[r].mask_dtype = mask.data_ptr(), _INT_WIDTH[mask.dtype]
        return arr, B

    def pool_roles(self, slot: int, blocks, stream: Optional[int] = None) -> None:
        """K1 over roles given as separate (input_ids [B, S_r], attention_mask [B, S_r] | None) blocks — what the
        reference's tokenizer call returns per encode (backend/model.py:43-48) — packed to int32 ids + prefix lengths
        on `stream` by the engine (tt_engine_pool_roles; needs HotPathEngine(pack_roles=True)).  0/1 masks only: a
        weight mask raises at the next read_stats()."""
        arr, B = self._role_array(blocks)
        call("tt_engine_pool_roles", self._h, slot, arr, len(blocks), B, self._stream() if stream is None else stream)

    def pack_roles(self, slot: int, blocks, stream: Optional[int] = None) -> int:
        """The packing half of pool_roles on `stream`; returns the packed rows' padded length for pool_packed()."""
        arr, B = self._role_array(blocks)
        S = C.c_int32(0)
        call("tt_engine_pack_roles", self._h, slot, arr, len(blocks), B, C.byref(S), self._stream() if stream is None else stream)
        return int(S.value)

    def pool_packed(self, slot: int, roles: int, B: int, S: int, stream: Optional[int] = None) -> None:
        """The gather half of pool_roles over the rows pack_roles() left in `slot` (order the two streams yourself)."""
        call("tt_engine_pool_packed", self._h, slot, roles, B, S, self._stream() if stream is None else stream)

    def pool_rows(self, slot: int, store_ids: torch.Tensor, store_lens: Optional[torch.Tensor], rows: torch.Tensor,
                  stream: Optional[int] = None) -> None:
        """K1 through a row indirection: text t of the batch is row rows[t] of the token store
        (store_ids [N, S] int32/int64, store_lens [N] int32 or None = full length); rows [roles, B] int32."""
        _req(store_ids, "store_ids", ndim=2)
        _req(rows, "rows", torch.int32, 2)
        if store_ids.dtype not in _INT_WIDTH or store_ids.device != self.device or rows.device != self.device:
            raise ValueError("pool_rows: store_ids must be an int32/int64 [N, S] tensor on the engine device")
        if rows.shape[0] not in (2, 3):
            raise ValueError("pool_rows: rows must be [roles, B] with roles 2 or 3")
        lp = None
        if store_lens is not None:
            _req(store_lens, "store_lens", torch.int32, 1)
            if store_lens.numel() != store_ids.shape[0] or store_lens.device != self.device:
                raise ValueError("pool_rows: store_lens must be int32 [N] on the engine device")
            lp = store_lens.data_ptr()
        call("tt_engine_pool_rows", self._h, slot, store_ids.data_ptr(), _INT_WIDTH[store_ids.dtype], lp,
             store_ids.shape[0], rows.data_ptr(), rows.shape[0], rows.shape[1], store_ids.shape[1],
             self._stream() if stream is None else stream)

    def set_backward(self, mode: str) -> None:
        """"auto" | "fork" | "fused": phase B's backward schedule (include/tt_hotpath.h, tt_engine_set_backward)."""
        call("tt_engine_set_backward", self._h, {"auto": _lib.TT_BWD_AUTO, "fork": _lib.TT_BWD_FORK,
                                                 "fused": _lib.TT_BWD_FUSED}[mode])
        self.backward_mode = mode

    def pool_cached(self, slot: int, cache: torch.Tensor, rows: torch.Tensor, stream: Optional[int] = None) -> None:
        """Phase A from a frozen-feature cache: pooled[slot] = cache[rows], rows [roles, B]."""
        _req(cache, "cache", torch.float32, 2)
        _req(rows, "rows", ndim=2)
        if cache.shape[1] != self.D or rows.dtype not in _INT_WIDTH or rows.shape[0] not in (2, 3):
            raise ValueError("pool_cached: cache must be [N, D] float32 and rows an int [roles, B] tensor")
        call("tt_engine_pool_cached", self._h, slot, cache.data_ptr(), cache.shape[0], rows.data_ptr(),
             _INT_WIDTH[rows.dtype], rows.shape[0], rows.shape[1], self._stream() if stream is None else stream)

    def fwd_bwd(self, slot: int, B: int, margin: float, grad_scale: Optional[float] = None,
                loss_denominator: Optional[float] = None, neg_index: Optional[torch.Tensor] = None) -> None:
        ni = None
        if neg_index is not None:
            _req(neg_index, "neg_index", torch.int32, 1)
            if neg_index.numel() != B or neg_index.device != self.device:
                raise ValueError("neg_index: expected an int32 [B] tensor on the engine device")
            ni = neg_index.data_ptr()
        call("tt_engine_fwd_bwd", self._h, slot, B, ni, margin, (1.0 / B) if grad_scale is None else grad_scale,
             float(B) if loss_denominator is None else loss_denominator, self._stream())

    def set_sharing(self, gathers_share_chip: bool) -> None:
        """Tell phase B whether token-id gathers are running beside it (picks the backward schedule)."""
        if getattr(self, "_sharing", None) != bool(gathers_share_chip):
            call("tt_engine_set_sharing", self._h, int(bool(gathers_share_chip)))
            self._sharing = bool(gathers_share_chip)

    def forward(self, slot: int, B: int, roles: int = 2) -> None:
        """Projection forward only; the embeddings are then in `embeddings(B, roles)`."""
        call("tt_engine_forward", self._h, slot, B, roles, self._stream())

    def backward(self, slot: int, B: int, roles: int = 2) -> None:
        """Parameter gradients from d loss / d embeddings written by the caller into `d_out(B, roles)`."""
        call("tt_engine_backward", self._h, slot, B, roles, self._stream())

    def add_loss(self, loss_value: torch.Tensor) -> None:
        _req(loss_value, "loss_value", torch.float32)
        call("tt_engine_add_loss", self._h, loss_value.data_ptr(), self._stream())

    def d_out(self, B: int, roles: int = 2) -> torch.Tensor:
        return self._view(_lib.load().tt_engine_dout(self._h), roles * B * self.P).view(roles, B, self.P)

    def reduce_grads(self, out: torch.Tensor, accumulate: bool = False) -> None:
        _req(out, "grad_out", torch.float32, 1)
        if out.numel() != self.n_params:
            raise ValueError("grad_out: wrong size")
        call("tt_engine_reduce_grads", self._h, out.data_ptr(), int(accumulate), self._stream())

    def k1_census(self, slot: int):
        """(workgroups admitted, compute units used, most candidates registered on one CU, workgroups planned) of the
        streaming gather last launched into `slot`; zeros when that gather did not stream.  Synchronises the current stream."""
        a, c, m, p = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        call("tt_engine_k1_census", self._h, int(slot), C.byref(a), C.byref(c), C.byref(m), C.byref(p), self._stream())
        return a.value, c.value, m.value, p.value

    def reduce_grads_range(self, out: torch.Tensor, offset: int, count: int, accumulate: bool = False) -> None:
        """out[offset : offset + count] (+)= slab sum of those parameters only (`out` is the whole flat bucket)."""
        _req(out, "grad_out", torch.float32, 1)
        if out.numel() != self.n_params:
            raise ValueError("grad_out: wrong size")
        call("tt_engine_reduce_grads_range", self._h, out.data_ptr(), int(offset), int(count), int(accumulate), self._stream())

    def set_dw2_event(self, event: Optional["torch.cuda.Event"]) -> None:
        """`event` is recorded inside fwd_bwd()/backward() as soon as the dW2 / db2 slabs of both towers are complete
        (tt_engine_set_dw2_event); None clears it."""
        if event is not None:
            event.record(torch.cuda.current_stream(self.device))  # torch creates the HIP event at its first record()
        self._dw2_event = event  # keep it alive while the engine holds its handle
        call("tt_engine_set_dw2_event", self._h, None if event is None else event.cuda_event)

    def set_loss_event(self, event) -> None:
        """`event` (_lib.StreamEvent or None) is recorded inside fwd_bwd() / backward() right after the loss kernel, on the
        caller's stream (tt_engine_set_loss_event) — in the fork backward it doubles as the fork point."""
        self._loss_event = event  # keep it alive while the engine holds its handle
        call("tt_engine_set_loss_event", self._h, None if event is None else event._h)

    def set_join_wait(self, event) -> bool:
        """One-shot (tt_engine_set_join_wait): the next fwd_bwd() / backward(), if it runs the fork backward, orders its join
        behind `event` (_lib.StreamEvent) as well.  True if the engine took it; False: wait for the event yourself."""
        took = C.c_int32(0)
        call("tt_engine_set_join_wait", self._h, event._h, C.byref(took))
        return bool(took.value)

    def arm_loss_event(self, armed: bool) -> None:
        """The fused backward records the loss event only while armed (tt_engine_arm_loss_event)."""
        call("tt_engine_arm_loss_event", self._h, 1 if armed else 0)

    def clear_join_wait(self) -> None:
        """Drop a pending one-shot join wait (a pipeline that is being abandoned)."""
        call("tt_engine_set_join_wait", self._h, None, None)

    def adam(self, step: int, lr: float, grads: Optional[torch.Tensor] = None, beta1=0.9, beta2=0.999, eps=1e-8) -> None:
        g = None
        if grads is not None:
            _req(grads, "grads", torch.float32, 1)
            g = grads.data_ptr()
        call("tt_engine_adam", self._h, g, lr, beta1, beta2, eps, step, self._stream())

    def adam_auto(self, lr: float, grads: Optional[torch.Tensor] = None, beta1=0.9, beta2=0.999, eps=1e-8) -> None:
        """Adam with the step counter on the device (graph-replayable); see `set_step`."""
        g = None
        if grads is not None:
            _req(grads, "grads", torch.float32, 1)
            g = grads.data_ptr()
        call("tt_engine_adam_auto", self._h, g, lr, beta1, beta2, eps, self._stream())

    def set_step(self, step: int) -> None:
        call("tt_engine_set_step", self._h, int(step), self._stream())

    def train_step(self, ids, mask=None, lens=None, *, margin: float, lr: float, step: int) -> None:
        a = self._ids_args(ids, mask, lens)
        if a[5] != 3:
            raise ValueError("train_step: ids must be [3, B, S]")
        call("tt_engine_train_step", self._h, a[0], a[1], a[2], a[3], a[4], a[6], a[7], margin, lr, step, self._stream())

    def read_stats(self, reset: bool = True) -> Tuple[float, int, float]:
        """(sum of batch losses, batches, last batch loss); synchronises the stream."""
        s, c, l = C.c_double(), C.c_int64(), C.c_float()
        try:
            call("tt_engine_read_stats", self._h, C.byref(s), C.byref(c), C.byref(l), int(reset), self._stream())
        except _lib.TTError as e:
            if e.code == -7:  # TT_ERR_UNSUPPORTED
                raise ValueError("an attention_mask with values other than 0 / 1 reached tt_engine_pool_roles (the packed "
                                 "prefix-length form cannot weigh tokens): the batches since the last read were pooled with "
                                 "every non-zero mask value as 1 — use pool() with the mask, or train_epoch(fused=False)") from e
            raise
        return s.value, c.value, l.value

    # ------------------------------------------------------------------ views
    def _view(self, ptr: int, numel: int) -> torch.Tensor:
        off = ptr - self.workspace.data_ptr()
        assert 0 <= off and off + 4 * numel <= self.workspace.numel()
        return self.workspace[off: off + 4 * numel].view(torch.float32)

    def pooled(self, slot: int, B: int, roles: int = 3) -> torch.Tensor:
        return self._view(_lib.load().tt_engine_pooled(self._h, slot), roles * B * self.D).view(roles, B, self.D)

    def embeddings(self, B: int, roles: int = 3) -> torch.Tensor:
        return self._view(_lib.load().tt_engine_embeddings(self._h), roles * B * self.P).view(roles, B, self.P)

    def last_loss(self) -> torch.Tensor:
        """The device scalar [1] the loss kernels leave each batch's loss in (a view: read it stream-ordered)."""
        return self._view(_lib.load().tt_engine_last_loss(self._h), 1)

    def grad_slabs(self) -> torch.Tensor:
        stride = (self.n_params + 3) // 4 * 4
        return self._view(_lib.load().tt_engine_grad_slab0(self._h), self.n_slabs * stride).view(self.n_slabs, stride)[:, : self.n_params]
