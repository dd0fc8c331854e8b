"""bench.py --path module / module-unfused: the drop-in surface a user of the reference gets by swapping the imports
 — `training.train_epoch(model, dataloader, TripletLoss, optimizer)` with the reference's
signature (backend/training.py:26-63), on the batch forms the reference's callers hand it:

* pre-tokenised (int64 input_ids, int64 attention_mask) per role — what the tokenizer returns (backend/model.py:43-48);
* `--strings`: lists of strings through a REAL `transformers.BertTokenizerFast` over a synthetic vocabulary of the
  table's size (every word one vocabulary token, texts of seq_len tokens with [CLS] / [SEP]) — epoch 1 pays the
  tokenizer once per text, epochs >= 2 run from the route's text store (two-towers-overlords_amd/textstore.py).

Every line has its own metric name; none is the BASELINE metric.  Imported by bench.py only."""
from __future__ import annotations

import contextlib
import os
import sys
import tempfile
import time

import numpy as np
import torch


def synthetic_tokenizer(V: int):
    """BertTokenizerFast from a vocab file written here (a local directory: nothing is fetched): 5 specials, then
    `w00005 .. w{V-1}` — every word of the synthetic texts is exactly one id of a V-row table."""
    import transformers

    d = tempfile.mkdtemp(prefix="tt_vocab_")
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"w{i:05d}" for i in range(5, V)]
    with open(os.path.join(d, "vocab.txt"), "w") as f:
        f.write("\n".join(vocab) + "\n")
    tok = transformers.BertTokenizerFast.from_pretrained(d)
    assert tok.vocab_size == V
    return tok


def synthetic_texts(n: int, words: int, V: int, seed: int, zipf: bool = False):
    """n texts of `words` vocabulary words each (-> words + 2 tokens)."""
    rng = np.random.default_rng(seed)
    if zipf:
        ids = np.exp(rng.uniform(0.0, np.log(V - 5), size=(n, words))).astype(np.int64) + 4
    else:
        ids = rng.integers(5, V, size=(n, words))
    names = np.array([f"w{i:05d}" for i in range(V)])
    return [" ".join(row) for row in names[ids]]


def _string_batches(args, B, S, V, tok):
    """A dataset of steps x B (query, positive) items and its batches in the form --strings names."""
    from two_towers_overlords_amd.data import TextColumn, TripletDataLoader

    n = args.steps * B
    t0 = time.perf_counter()
    if args.ragged:  # MS-MARCO-like lengths (bench_eval.MarcoLike): queries of 6-16 tokens, passages of 32-128
        rng = np.random.default_rng(99)

        def ragged(lo, hi, seed):
            words = rng.integers(lo, hi + 1, size=n)
            by_len = {int(w): iter(synthetic_texts(int((words == w).sum()), int(w), V, seed + int(w), args.zipf)) for w in np.unique(words)}
            return [next(by_len[int(w)]) for w in words]

        queries, positives = ragged(4, 14, 1234), ragged(30, S - 2, 4321)
    else:
        queries = synthetic_texts(n, S - 2, V, 1234, args.zipf)
        positives = synthetic_texts(n, S - 2, V, 4321, args.zipf)
    items = [{"query_id": i, "query": q, "positive": p} for i, (q, p) in enumerate(zip(queries, positives))]
    build_s = time.perf_counter() - t0

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return n

        def __getitem__(self, i):
            return items[i]

    ds = DS()
    if args.strings == "loader":
        return TripletDataLoader(ds, batch_size=B, num_workers=0), n, build_s
    rng = np.random.default_rng(7)
    batches = []
    for s in range(0, n, B):
        rows = np.arange(s, s + B)
        neg = (np.arange(B) + rng.integers(1, B, size=B)) % B  # an in-batch positive of ANOTHER item (backend/data.py:113-137)
        q, p, ng = [queries[i] for i in rows], [positives[i] for i in rows], [positives[s + j] for j in neg]
        if args.strings == "columns":
            batches.append((TextColumn(q, ds, "query", rows), TextColumn(p, ds, "positive", rows),
                            TextColumn(ng, ds, "positive", s + neg)))
        else:
            batches.append((q, p, ng))
    return batches, n, build_s


def run_module_path(args, device, CONFIGS, make_ids) -> dict:
    import two_towers_overlords_amd as pkg
    from two_towers_overlords_amd.model import EmbeddingOnlyEncoder, TripletLoss, TwoTowersModel
    from two_towers_overlords_amd.training import FusedAdam, train_epoch

    pkg.load()
    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    V, D, P, B, S = cfg["V"], cfg["D"], cfg["P"], cfg["B"], cfg["S"]
    gt = torch.Generator().manual_seed(1)
    table = torch.randn(V, D, generator=gt) * 0.02
    torch.manual_seed(2)
    tok = synthetic_tokenizer(V) if args.strings else (lambda *a, **k: None)
    model = TwoTowersModel(projection_dim=P, tokenizer=tok, pretrained_model=EmbeddingOnlyEncoder(table),
                           encoder_mode="embed").to(device)
    crit = TripletLoss(margin=0.3)
    # backend/training.py:436: `torch.optim.Adam(model.parameters(), lr=learning_rate)` — the reference's literal optimizer
    opt = (torch.optim.Adam if args.optimizer == "torch" else FusedAdam)(model.parameters(), lr=1e-3)
    fused = None if args.path == "module" else False
    opt_name = "torch.optim.Adam" if args.optimizer == "torch" else "FusedAdam"
    common = {"unit": "triplets/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "roofline": None}
    config = {"workload": cfg["label"], "path": args.path, "optimizer": opt_name, "encoder_mode": "embed", "vocab": V,
              "embed_dim": D, "proj_dim": P, "seq_len": S, "batch_per_gpu": B, "global_batch": B, "parallelism": "dp1"}
    route = lambda: getattr(opt, "_tt_route", None)

    if args.strings:
        batches, n_items, build_s = _string_batches(args, B, S, V, tok)
        epochs = []
        with contextlib.redirect_stdout(sys.stderr):  # train_epoch prints every 10th batch's loss, like the reference
            for e in range(max(args.epochs or 3, 2)):
                st0 = route().text_store.stats() if route() is not None and route().text_store is not None else None
                torch.cuda.synchronize(device)
                prof = None
                if args.debug_hostprof and e == max(args.epochs or 3, 2) - 1:  # cProfile of the last (warm) epoch's host side
                    import cProfile

                    prof = cProfile.Profile()
                    prof.enable()
                t0 = time.perf_counter()
                avg = train_epoch(model, batches, crit, opt, log_wandb=False, fused=fused, text_cache=(None if args.text_cache else False))
                host_dt = time.perf_counter() - t0  # the host is done enqueueing; the device may still be working
                torch.cuda.synchronize(device)
                dt = time.perf_counter() - t0
                if prof is not None:
                    import pstats

                    prof.disable()
                    pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(30)
                st1 = route().text_store.stats() if route() is not None and route().text_store is not None else None
                tok_s = (st1["tokenize_s"] - (st0["tokenize_s"] if st0 else 0.0)) if st1 else None
                epochs.append({"epoch": e + 1, "seconds": dt, "triplets_per_s": n_items / dt, "us_per_batch": dt / args.steps * 1e6,
                               "avg_loss": avg, "tokenizer_s": tok_s, "host_us_per_batch": host_dt / args.steps * 1e6,
                               "tokenizer_share": (tok_s / dt) if tok_s is not None else None,
                               "texts_tokenized": (st1["texts_tokenized"] - (st0["texts_tokenized"] if st0 else 0)) if st1 else None})
        warm = epochs[1:]
        best = min(warm, key=lambda r: r["seconds"])
        return {**common,
                "metric": (f"triplets/sec, drop-in train_epoch(model, STRING batches, TripletLoss, {opt_name}) through a real "
                           f"BertTokenizerFast, batches as {args.strings}; value = the best epoch >= 2 "
                           + ("(texts are rows of the route's device-resident text store; no tokenizer call)" if args.text_cache else
                              "(text_cache=False: three tokenizer calls per batch, as backend/model.py:43-48)")
                           + "; epoch 1 (tokenises every text once) in `epochs`; NOT the BASELINE metric"),
                "value": best["triplets_per_s"], "ms_per_step": best["us_per_batch"] / 1e3,
                "config": {**config, "strings": args.strings, "text_cache": bool(args.text_cache), "items": n_items,
                           "texts": 2 * n_items, "tokens_per_text": ("queries 6-16, passages 32-%d" % S) if args.ragged else S,
                           "ids": "zipf" if args.zipf else "uniform",
                           "tokenizer": "transformers.BertTokenizerFast over a synthetic vocab.txt of V words",
                           "host_threads": os.cpu_count(), "dataset_build_s": build_s},
                "epochs": epochs, "first_epoch": epochs[0],
                "text_store": route().text_store.stats() if route() is not None and route().text_store is not None else None,
                "routed": route() is not None,
                "calibration": route().trainer.calibration if route() is not None else None}

    ids = make_ids(V, B, S, 8, seed=1234, device=device, zipf=args.zipf)
    ones = torch.ones(B, S, dtype=torch.int64, device=device)
    # what the tokenizer hands over (backend/model.py:43-48): int64 input_ids and attention_mask per encode call
    distinct = [tuple((b[r].long(), ones) for r in range(3)) for b in ids]
    seq = lambda n: [distinct[i % len(distinct)] for i in range(n)]
    with contextlib.redirect_stdout(sys.stderr):
        train_epoch(model, seq(max(args.warmup, 1)), crit, opt, log_wandb=False, fused=fused)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        avg = train_epoch(model, seq(args.steps), crit, opt, log_wandb=False, fused=fused)
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
    return {**common,
            "metric": (f"triplets/sec, drop-in train_epoch(model, batches, TripletLoss, {opt_name}) — batch body routed to the fused "
                       "engine; pipeline fill, drain and the loss reads inside the timed epoch; NOT the BASELINE metric"
                       if fused is None else
                       f"triplets/sec, drop-in MODULE path, fused=False: training.train_epoch over TwoTowersModel.encode_* + TripletLoss "
                       f"+ {opt_name} (autograd Functions over the same HIP kernels); NOT the BASELINE metric"),
            "value": args.steps * B / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
            "config": {**config,
                       "ids": "int64 input_ids + int64 attention_mask [B,S] per encode call, " + ("zipf" if args.zipf else "uniform")
                              + ", full length",
                       "launches_per_step": ("the fused engine's (one gather for the three roles two batches ahead on a side stream; "
                                             "2 forward GEMM launches, loss, 2-3 backward launches, Adam) + the engine-side role "
                                             "packing pass" if fused is None else
                                             "one gather for the three roles, projection fwd x2 per tower, triplet fwd+bwd, projection "
                                             "bwd x3 + one slab reduction per tower, the optimizer's launches, ~5 torch ops")},
            "avg_loss": avg, "routed": route() is not None,
            "calibration": route().trainer.calibration if route() is not None else None}
