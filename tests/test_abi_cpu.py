"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/tt_hotpath.h declares; argument errors are caught
on the host; nothing falls back to a CPU implementation."""
import ctypes
import os
import re

import pytest
import torch

import two_towers_overlords_amd as pkg
from two_towers_overlords_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "tt_hotpath.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tt_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported_and_bound():
    if not os.path.exists(_lib.LIB_PATH):
        pkg.build()
    lib = pkg.load()
    syms = declared_symbols()
    assert len(syms) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/tt_hotpath.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.tt_abi_version() == _lib.ABI_VERSION
    assert lib.tt_status_string(0) == b"ok"
    assert b"aligned" in lib.tt_status_string(-4)


def test_host_side_argument_errors_no_gpu_needed():
    lib = pkg.load()
    # null pointers / bad shapes / bad dtypes are rejected before any launch
    assert lib.tt_gather_pool_norm_f32(None, None, 8, None, 8, None, None, 1, 1, 4, 10, None, None) == -1
    assert lib.tt_gather_pool_norm_f32(16, 16, 8, None, 8, None, 16, 1, 0, 4, 10, None, None) == -2
    assert lib.tt_gather_pool_norm_f32(16, 16, 3, None, 8, None, 16, 1, 1, 4, 10, None, None) == -3
    assert lib.tt_adam_f32(16, 16, 1, 0, 16, 16, 8, 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, None) == -2  # step must be >= 1
    assert lib.tt_adam_f32(20, 16, 1, 0, 16, 16, 8, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None) == -4  # alignment
    assert lib.tt_engine_destroy(None) == -6
    assert lib.tt_proj_bwd_workspace_bytes(2048, 384, 384) > 0


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing on the host."""
    t = torch.zeros(4, 4)
    ids = torch.zeros(2, 2, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gather_pool_norm(t, ids)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.triplet_fwd_bwd(t, t, t, 0.3)


def test_product_package_never_imports_oracle():
    pkg_dir = os.path.join(ROOT, "two-towers-overlords_amd")
    for dp, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
