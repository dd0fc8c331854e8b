"""ctypes binding of libtt_hotpath.so (the C ABI declared in include/tt_hotpath.h).

The product path has no fallback: if the library is missing or fails to load,
every op raises.  `build()` compiles it in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# TT_HOTPATH_LIB selects another build of the same ABI (A/B runs of two library versions on one box)
LIB_PATH = os.environ.get("TT_HOTPATH_LIB") or os.path.join(_HERE, "libtt_hotpath.so")
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

TT_I32, TT_I64 = 4, 8
TT_BWD_AUTO, TT_BWD_FORK, TT_BWD_FUSED = 0, 1, 2
TT_K1_SHARE_AUTO = -1  # tt_engine_config.k1_blocks_per_cu: pair the K1 form with the projection tile
TT_K1_AUTO, TT_K1_PER_TEXT, TT_K1_ROW_SLICES, TT_K1_COLUMN_SLICES, TT_K1_STREAM, TT_K1_BINNED = 0, 1, 2, 3, 4, 5
K1_FORMS = {"auto": TT_K1_AUTO, "per_text": TT_K1_PER_TEXT, "rows": TT_K1_ROW_SLICES, "columns": TT_K1_COLUMN_SLICES,
            "stream": TT_K1_STREAM, "binned": TT_K1_BINNED}
ABI_VERSION = 4

_lib = None
_lock = threading.Lock()


class TTError(RuntimeError):
    """A C-ABI call returned a non-zero tt_status / hipError_t."""

    def __init__(self, fn: str, code: int, msg: str):
        super().__init__(f"{fn} failed with status {code}: {msg}")
        self.code = code


class TTEngineConfig(C.Structure):
    _fields_ = [
        ("V", C.c_int64),
        ("D", C.c_int32),
        ("P", C.c_int32),
        ("max_B", C.c_int32),
        ("max_S", C.c_int32),
        ("device", C.c_int32),
        ("k1_blocks_per_cu", C.c_int32),
        ("grad_slabs", C.c_int32),
        ("k1_form", C.c_int32),
        ("pack_roles", C.c_int32),
    ]


class TTRoleTokens(C.Structure):
    """tt_role_tokens: one role's [B, S] ids (+ attention mask) as the tokenizer returned them (device pointers)."""

    _fields_ = [("ids", C.c_void_p), ("mask", C.c_void_p), ("ids_dtype", C.c_int32), ("mask_dtype", C.c_int32),
                ("S", C.c_int32), ("reserved", C.c_int32)]


class TTAdamTensor(C.Structure):
    """tt_adam_tensor: one tensor of a multi-tensor Adam launch (device pointers as integers)."""

    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64)]


class TTTuning(C.Structure):
    """tt_tuning (include/tt_hotpath.h): every field 0 = auto.  Speed only, never results."""

    _fields_ = [
        ("k1_form", C.c_int32),
        ("k1_stream_wgs", C.c_int32),
        ("k1_stream_rows", C.c_int32),
        ("k1_stream_offer", C.c_int32),
        ("k1_share_lds_kb", C.c_int32),
        ("proj_tile", C.c_int32),
        ("proj_bk", C.c_int32),
        ("tn_tile", C.c_int32),
        ("xcd_order", C.c_int32),
        ("backward", C.c_int32),
        ("event_scope", C.c_int32),
        ("reserved", C.c_int32 * 5),
    ]


TUNING_FIELDS = tuple(n for n, _ in TTTuning._fields_ if n != "reserved")


def make_tuning(tuning=None):
    """None | dict | TTTuning -> TTTuning or None.  Dict values may name a K1 form ("stream", "columns", ...) or a
    backward schedule ("fork" / "fused"); unknown keys raise."""
    if tuning is None or isinstance(tuning, TTTuning):
        return tuning
    t = TTTuning()
    for k, v in dict(tuning).items():
        if k not in TUNING_FIELDS:
            raise KeyError(f"tt_tuning has no field {k!r}; fields: {', '.join(TUNING_FIELDS)}")
        if k == "k1_form" and isinstance(v, str):
            v = K1_FORMS[v]
        if k == "backward" and isinstance(v, str):
            v = {"auto": TT_BWD_AUTO, "fork": TT_BWD_FORK, "fused": TT_BWD_FUSED}[v]
        setattr(t, k, int(v))
    return t


def tuning_ref(tuning):
    t = make_tuning(tuning)
    return None if t is None else C.byref(t)


_tp = C.POINTER(TTTuning)
_vp, _i, _i32, _i64, _f, _d, _sz = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/tt_hotpath.h one to one
SIGNATURES = {
    "tt_abi_version": (_i, []),
    "tt_status_string": (C.c_char_p, [_i]),
    "tt_gather_pool_norm_f32": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i64, _i, _i, _i64, _vp, _vp]),
    "tt_gather_pool_sliced_workspace_bytes": (_sz, [_i64, _i]),
    "tt_k1_prefers_sliced": (_i, [_i64, _i, _i64, _i]),
    "tt_k1_sliced_form": (_i, [_i64, _i, _i]),
    "tt_gather_pool_norm_sliced_f32": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i64, _i, _i, _i64, _vp, _vp, _sz, _i, _vp]),
    "tt_gather_pool_binned_workspace_bytes": (_sz, [_i64, _i, _i]),
    "tt_k1_binned_supported": (_i, [_i, _i, _vp, _vp]),
    "tt_k1_prefers_binned": (_i, [_i64, _i, _i64, _i]),
    "tt_gather_pool_norm_binned_f32": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i64, _i, _i, _i64, _vp, _vp, _sz, _vp]),
    "tt_gather_pool_stream_workspace_bytes": (_sz, []),
    "tt_k1_stream_supported": (_i, [_i, _vp, _vp]),
    "tt_k1_prefers_stream": (_i, [_i64, _i, _i64, _i]),
    "tt_gather_pool_norm_stream_f32": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i64, _i, _i, _i64, _vp, _vp, _sz, _i, _tp, _vp]),
    "tt_gather_pool_stream_census": (_i, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), _vp]),
    "tt_pool_norm_dense_f32": (_i, [_vp, _vp, _i, _vp, _vp, _i64, _i, _i, _vp]),
    "tt_embed_gather_f32": (_i, [_vp, _vp, _i, _vp, _i64, _i, _i64, _vp, _vp]),
    "tt_proj_fwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _tp, _vp]),
    "tt_proj_bwd_workspace_bytes": (_sz, [_i64, _i, _i]),
    "tt_proj_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i64, _i, _i, _tp, _vp]),
    "tt_triplet_fwd_bwd_f32": (_i, [_vp, _vp, _vp, _f, _f, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tt_adam_f32": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _i64, _d, _d, _d, _d, _i64, _f, _vp]),
    "tt_adam_multi_f32": (_i, [C.POINTER(TTAdamTensor), _i, _d, _d, _d, _d, _i64, _vp]),
    "tt_cosine_distance_f32": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "tt_cosine_distance_bwd_f32": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "tt_reduce_slabs_f32": (_i, [_vp, _i, _i64, _vp, _i64, _i, _vp]),
    "tt_inbatch_workspace_bytes": (_sz, [_i64, _i64, _i]),
    "tt_inbatch_triplet_fwd_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _f, _f, _i64, _i64, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "tt_cosine_scores_workspace_bytes": (_sz, [_i64, _i64, _i]),
    "tt_cosine_scores_f32": (_i, [_vp, _vp, _vp, _i64, _i64, _i, _vp, _sz, _vp]),
    "tt_ndcg_at_k_f32": (_i, [_vp, _vp, _vp, _i64, _i64, C.POINTER(_i32), _i, _vp, _vp, _vp]),
    "tt_engine_workspace_bytes": (_sz, [C.POINTER(TTEngineConfig)]),
    "tt_engine_create": (_i, [C.POINTER(TTEngineConfig), _vp, _sz, _vp, C.POINTER(_vp)]),
    "tt_engine_destroy": (_i, [_vp]),
    "tt_engine_set_sharing": (_i, [_vp, _i]),
    "tt_engine_set_backward": (_i, [_vp, _i]),
    "tt_engine_set_tuning": (_i, [_vp, _tp]),
    "tt_engine_get_tuning": (_i, [_vp, _tp]),
    "tt_engine_pool_rows": (_i, [_vp, _i, _vp, _i, _vp, _i64, _vp, _i32, _i32, _i32, _vp]),
    "tt_engine_param_count": (_i64, [_vp]),
    "tt_engine_grad_slabs": (_i, [_vp]),
    "tt_engine_bind": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "tt_engine_pool": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i32, _i32, _i32, _vp]),
    "tt_engine_pool_cached": (_i, [_vp, _i, _vp, _i64, _vp, _i, _i32, _i32, _vp]),
    "tt_engine_pool_roles": (_i, [_vp, _i, C.POINTER(TTRoleTokens), _i32, _i32, _vp]),
    "tt_engine_pack_roles": (_i, [_vp, _i, C.POINTER(TTRoleTokens), _i32, _i32, C.POINTER(_i32), _vp]),
    "tt_engine_pool_packed": (_i, [_vp, _i, _i32, _i32, _i32, _vp]),
    "tt_engine_fwd_bwd": (_i, [_vp, _i, _i32, _vp, _f, _f, _f, _vp]),
    "tt_engine_forward": (_i, [_vp, _i, _i32, _i32, _vp]),
    "tt_engine_backward": (_i, [_vp, _i, _i32, _i32, _vp]),
    "tt_engine_dout": (_vp, [_vp]),
    "tt_engine_add_loss": (_i, [_vp, _vp, _vp]),
    "tt_engine_reduce_grads": (_i, [_vp, _vp, _i, _vp]),
    "tt_engine_k1_census": (_i, [_vp, _i, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), _vp]),
    "tt_engine_reduce_grads_range": (_i, [_vp, _vp, _i64, _i64, _i, _vp]),
    "tt_engine_set_dw2_event": (_i, [_vp, _vp]),
    "tt_engine_set_loss_event": (_i, [_vp, _vp]),
    "tt_engine_arm_loss_event": (_i, [_vp, _i32]),
    "tt_engine_set_join_wait": (_i, [_vp, _vp, C.POINTER(C.c_int32)]),
    "tt_event_create": (_i, [_i, C.POINTER(_vp)]),
    "tt_event_destroy": (_i, [_vp]),
    "tt_event_record": (_i, [_vp, _vp]),
    "tt_stream_wait_event": (_i, [_vp, _vp]),
    "tt_event_done": (_i, [_vp]),
    "tt_engine_adam": (_i, [_vp, _vp, _d, _d, _d, _d, _i64, _vp]),
    "tt_engine_adam_auto": (_i, [_vp, _vp, _d, _d, _d, _d, _vp]),
    "tt_engine_set_step": (_i, [_vp, _i64, _vp]),
    "tt_engine_train_step": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i32, _i32, _f, _d, _i64, _vp]),
    "tt_engine_pooled": (_vp, [_vp, _i]),
    "tt_engine_embeddings": (_vp, [_vp]),
    "tt_engine_grad_slab0": (_vp, [_vp]),
    "tt_engine_last_loss": (_vp, [_vp]),
    "tt_engine_read_stats": (_i, [_vp, C.POINTER(_d), C.POINTER(_i64), C.POINTER(_f), _i, _vp]),
}


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip -> libtt_hotpath.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libtt_hotpath.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout)
    return LIB_PATH


def load():
    """Load the library (once) and attach the signatures.  Raises if it is absent —
    there is deliberately no CPU / PyTorch fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950). The HIP library is the only implementation of this path."
            )
        # torch ships its own libamdhip64; it must be in the process first so that this
        # library binds to the SAME HIP runtime (streams and pointers are shared with torch).
        import torch  # noqa: F401

        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        v = lib.tt_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f"libtt_hotpath ABI {v} != expected {ABI_VERSION}; rebuild")
        _lib = lib
    return _lib


def check(fn: str, rc: int) -> None:
    if rc != 0:
        raise TTError(fn, rc, load().tt_status_string(rc).decode())


def call(fn: str, *args) -> None:
    check(fn, getattr(load(), fn)(*args))


class StreamEvent:
    """A HIP event for ordering streams of ONE device against each other (tt_event_* in include/tt_hotpath.h); scope 1
    (default here): no system-scope fence when it is recorded.  Streams are torch.cuda.Stream objects."""

    def __init__(self, scope: int = 1):
        h = C.c_void_p()
        call("tt_event_create", int(scope), C.byref(h))
        self._h = h
        self.recorded = False

    def record(self, stream) -> None:
        call("tt_event_record", self._h, stream.cuda_stream)
        self.recorded = True

    def wait(self, stream) -> None:
        """`stream` waits for the work recorded before this event."""
        if self.recorded:
            call("tt_stream_wait_event", stream.cuda_stream, self._h)

    def query(self) -> bool:
        if not self.recorded:
            return True
        r = load().tt_event_done(self._h)
        if r < 0:
            check("tt_event_done", r)
        return r == 1

    def __del__(self):
        try:
            if self._h and _lib is not None:
                _lib.tt_event_destroy(self._h)
        except Exception:
            pass
