"""two-towers-overlords_amd — MI355X (gfx950) native hot path of the two-tower
triplet trainer: token-embedding gather + masked mean-pool + L2-normalise,
per-tower Linear/ReLU/Linear projection (fp32 MFMA), cosine-distance triplet
loss, Adam — hand-written HIP kernels behind a C ABI (include/tt_hotpath.h),
under a Python surface that mirrors the reference's backend/model.py and
backend/training.py.

Import as `two_towers_overlords_amd` (the directory name carries a hyphen; the
root-level module of that name forwards here)."""

from . import _lib  # noqa: F401
from ._lib import TTError, build, load  # noqa: F401

__all__ = ["build", "load", "TTError"]
