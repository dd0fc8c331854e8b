"""Kernel variants that a policy picks inside the engine, forced here for every shape of the parity suite.
The kernels the engine pairs with the streaming gather (BASELINE.json configs[3]): 128x128 projection tiles with
K-step 16 (NT / NN / split-K TN) in the XCD-aware workgroup order.  They are selected by policy inside the engine;
here the A/B environment knobs force them for EVERY shape the parity suite knows (the knobs are read once per
process, hence the child process), so odd shapes, edge tiles and the scalar path go through them as well."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", [
    {"TT_TILE": "22", "TT_BK": "16", "TT_TN_TILE": "22", "TT_XCD": "1"},
    {"TT_XCD": "1"},                      # the XCD-aware order over the default tiles (64x64, 128x64, 128x128 K-step 32)
    {"TT_TILE": "21", "TT_XCD": "1"},
    {"TT_G16": "1", "TT_G16_FORCE": "1", "TT_G16_NN": "1"},  # 96x96 tiles on the 16x16x4 MFMA (gemm16.hip) for every forward / dH product
    # split-bf16 products (three bf16 terms per operand, six MFMA products: fp32-accurate at 3/8 of the fp32 MFMA's
    # cycles) — an opt-in experiment (TT_BF3=1), held to the SAME 1e-5 / 1e-6 bounds as the fp32 kernels
    {"TT_BF3": "1"},
    {"TT_BF3": "1", "TT_TILE": "22", "TT_BK": "16", "TT_TN_TILE": "22"},
    {"TT_BF3": "2"},                      # the same products with the split done once per workgroup at staging time
    {"TT_BF3": "2", "TT_TILE": "22", "TT_TN_TILE": "22"},
    {"TT_BF3": "2", "TT_TILE": "21"},
    {"TT_BF3": "3"},
    {"TT_INBATCH_TILE": "2"},             # 128x128 tiles (two epilogue passes) for the in-batch similarity product                      # A/B mode: staged split for kNT only, fragment split for kNN / kTN
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_parity_suite_under_the_stream_policy_kernels(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu",
                        "-k", "projection or engine_matches_reference_golden or engine_vs_oracle or awkward or neg_index or "
                              "feature_cache or metric_size or inbatch"],
                       capture_output=True, text=True, env=e, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
