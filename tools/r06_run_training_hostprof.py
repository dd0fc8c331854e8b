#!/usr/bin/env python3
"""cProfile of one run_training(...) call's host side (embed mode, 100,000 pairs x 15 epochs): where the wall clock that is
neither the tokenizer nor the device goes.   python tools/r06_run_training_hostprof.py [pairs]"""
import cProfile
import contextlib
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import two_towers_overlords_amd as pkg  # noqa: E402
from bench_eval import MarcoLike  # noqa: E402
from bench_module import synthetic_tokenizer  # noqa: E402
from two_towers_overlords_amd.model import EmbeddingOnlyEncoder  # noqa: E402
from two_towers_overlords_amd.training import run_training  # noqa: E402

pkg.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
V, D = 30522, 384
tok = synthetic_tokenizer(V)
train, val, test = MarcoLike(N, V, 21), MarcoLike(1000, V, 22), MarcoLike(10000, V, 23)
tok(["w00010"] * 64)  # the tokenizer's thread pool spins up at its first batched call
torch.zeros(1).cuda()
pr = cProfile.Profile()
with contextlib.redirect_stdout(sys.stderr):
    pr.enable()
    run_training(num_epochs=15, batch_size=1024, learning_rate=1e-4, max_samples=N, projection_dim=512, margin=0.3, use_wandb=False,
                 accumulation_steps=1, train_dataset=train, val_dataset=val, test_dataset=test, tokenizer=tok,
                 pretrained_model=EmbeddingOnlyEncoder(torch.randn(V, D) * 0.02), encoder_mode="embed", seed=5)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
