#!/usr/bin/env python3
"""cProfile of evaluate_model's host side at the comprehensive shape (bench_eval.py's data): where the 84 ms of
`encode` that are not the tokenizer go.  python tools/r06_eval_hostprof.py"""
import cProfile
import contextlib
import os
import pstats
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import two_towers_overlords_amd as pkg  # noqa: E402
from bench_eval import MarcoLike, timed_tokenizer  # noqa: E402
from bench_module import synthetic_tokenizer  # noqa: E402
from two_towers_overlords_amd.evaluation import evaluate_model  # noqa: E402
from two_towers_overlords_amd.model import EmbeddingOnlyEncoder, TwoTowersModel  # noqa: E402

pkg.load()
V, D, P = 30522, 384, 384
table = torch.randn(V, D) * 0.02
tok = timed_tokenizer(synthetic_tokenizer(V))
model = TwoTowersModel(projection_dim=P, tokenizer=tok, pretrained_model=EmbeddingOnlyEncoder(table), encoder_mode="embed").cuda()
test = MarcoLike(10000, V, 12)
kw = dict(sample_size=len(test), min_query_groups=200, candidate_pool_size=-1, comprehensive=True, batch_size=1024)
with contextlib.redirect_stdout(sys.stderr):
    for rep in range(3):
        random.seed(rep)
        pr = cProfile.Profile()
        pr.enable()
        evaluate_model(model, test, **kw)
        torch.cuda.synchronize()
        pr.disable()
pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(25)
