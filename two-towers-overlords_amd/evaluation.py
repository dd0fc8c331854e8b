"""Host-side mirror of `evaluate_model` (backend/training.py:136-380) with the arithmetic on
the device.

Same signature, same sampling / grouping / candidate-pool construction (same `random` call
sequence), same returned values.  What changes is where the numbers are computed:

    reference (per query, in a Python loop)                  here (all queries at once)
    -------------------------------------------------------  -----------------------------------------
    model.encode_queries([query])            B = 1            one encode of all selected queries
    model.encode_documents_batched(pool)     per query        one encode of the UNIQUE candidate texts
    torch.cosine_similarity(q.cpu(), d.cpu())                 tt_cosine_scores_f32  (MFMA GEMM, [Q, N])
    sklearn ndcg_score(k=10 / 5 / 1)                          tt_ndcg_at_k_f32      (tie-averaged, on device)

A query's candidate pool is a row mask over the unique documents, so per-query pools cost
nothing extra.  wandb logging and the sample print-outs are reproduced only as far as they do
not need the host copies the reference makes.
"""
from __future__ import annotations

import contextlib
import dataclasses
import random
import time
from typing import Dict, List, Optional, Set, Tuple, Union

import numpy as np
import torch

from . import ops
from .model import TwoTowersModel


@dataclasses.dataclass
class QueryGroup:
    """One evaluation query and the passages the sampled items list for it."""

    query: str
    relevant: Set[str] = dataclasses.field(default_factory=set)


def sample_query_groups(dataset, sample_size: int, min_query_groups: int, rng=random) -> Tuple[List, Dict]:
    """Draw evaluation queries the way backend/training.py:184-207 does — the draws have to be the reference's
    (`rng.sample(range(len(dataset)), k)` with k = sample_size, 2*sample_size, ... at most ten times, until
    `min_query_groups` distinct query ids have been seen), or `tests/golden/eval.npz` would not be reproduced.
    Returns (the query ids with the most relevant passages first, cut to min_query_groups; {query_id: QueryGroup})."""
    groups: Dict = {}
    for attempt in range(1, 11):
        if len(groups) >= min_query_groups:
            break
        for i in rng.sample(range(len(dataset)), min(sample_size * attempt, len(dataset))):
            item = dataset[i]
            groups.setdefault(item["query_id"], QueryGroup(item["query"])).relevant.add(item["positive"])
    ranked = sorted(groups, key=lambda qid: len(groups[qid].relevant), reverse=True)  # stable: ties keep first-seen order
    return ranked[:min_query_groups], groups


def candidate_pools(order: List, groups: Dict, candidate_pool_size: int, rng=random) -> List[List[str]]:
    """Per query: its relevant passages first, then the other selected queries' passages it does not share, thinned
    to `candidate_pool_size` by `rng.sample` when there are more (backend/training.py:249-267)."""
    pools = []
    for qid in order:
        relevant = groups[qid].relevant
        others: Set[str] = set()
        for other in order:
            if other != qid:
                others.update(d for d in groups[other].relevant if d not in relevant)
        distractors = list(others)
        if len(distractors) > candidate_pool_size:
            distractors = rng.sample(distractors, candidate_pool_size)
        pools.append(list(relevant) + distractors)
    return pools


def evaluate_model(
    model: TwoTowersModel,
    dataset,
    sample_size: int = 200,
    min_query_groups: int = 20,
    candidate_pool_size: int = 100,
    comprehensive: bool = False,
    log_wandb: bool = False,
    wandb_prefix: str = "final_",
    batch_size: int = 1024,
    *,
    timings: Optional[dict] = None,
) -> Union[float, dict]:
    """NDCG@{10,5,1} over sampled query groups.  Mirrors backend/training.py:136-380.
    `timings` (additive): a dict that receives the wall seconds of each stage (the device is synchronised at every stage
    boundary, so the stages add up to the call; leave it None outside measurements) plus the evaluation's shape."""

    @contextlib.contextmanager
    def stage(name):
        if timings is None:
            yield
            return
        t0 = time.perf_counter()
        yield
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        timings[name] = timings.get(name, 0.0) + time.perf_counter() - t0

    print(f"Sampling documents for evaluation (initial sample size: {sample_size})...")
    model.eval()
    with stage("sample_s"):
        query_ids, groups = sample_query_groups(dataset, sample_size, min_query_groups)
    total_relevant_docs = sum(len(groups[qid].relevant) for qid in query_ids)
    avg_relevant_per_query = total_relevant_docs / len(query_ids)
    print(f"Selected {len(query_ids)} queries for evaluation")

    # candidate pools as (unique document list, per-query membership mask): -1 = every unique passage of the dataset
    with stage("pools_s"):
        if candidate_pool_size == -1:
            docs = list(dataset.get_unique_passages())
            pools = None
        else:
            pools = candidate_pools(query_ids, groups, candidate_pool_size)
            docs = list(dict.fromkeys(d for pool in pools for d in pool))
        doc_index = {d: j for j, d in enumerate(docs)}
        Q, N = len(query_ids), len(docs)

        relevance = np.zeros((Q, N), dtype=np.float32)
        valid = None if pools is None else np.zeros((Q, N), dtype=np.uint8)
        for i, query_id in enumerate(query_ids):
            for d in groups[query_id].relevant:
                j = doc_index.get(d)
                if j is not None:
                    relevance[i, j] = 1.0
            if pools is not None:
                valid[i, [doc_index[d] for d in pools[i]]] = 1

    # -- embeddings through the hot path, similarities and NDCG on the device
    with torch.no_grad():
        with stage("encode_s"):  # tokenizer + K1 + K2 (the tokenizer's share: TwoTowersModel's towers time it, see bench_eval.py)
            query_embeds = model.encode_queries([groups[qid].query for qid in query_ids])
            doc_embeds = model.encode_documents_batched(docs, batch_size=batch_size)
        with stage("scores_s"):
            scores = ops.cosine_scores(query_embeds.contiguous(), doc_embeds.contiguous())
        with stage("ndcg_s"):
            dev = scores.device
            ndcg = ops.ndcg_at_k(
                scores,
                torch.from_numpy(relevance).to(dev),
                ks=(10, 5, 1),
                valid=None if valid is None else torch.from_numpy(valid).to(dev),
            ).double().cpu().numpy()
    if timings is not None:
        timings.update(Q=Q, N=N, P=int(query_embeds.shape[1]))

    ndcg_10_scores = ndcg[:, 0]
    mean_ndcg_10 = float(np.mean(ndcg_10_scores))
    if not comprehensive:
        print(f"\nMean NDCG@10 across {len(query_ids)} queries: {mean_ndcg_10:.4f}")
        return mean_ndcg_10

    results = {
        f"{wandb_prefix}ndcg_10": mean_ndcg_10,
        f"{wandb_prefix}ndcg_5": float(np.mean(ndcg[:, 1])),
        f"{wandb_prefix}ndcg_1": float(np.mean(ndcg[:, 2])),
        f"{wandb_prefix}ndcg_10_std": float(np.std(ndcg_10_scores)),
        f"{wandb_prefix}queries_evaluated": len(query_ids),
        f"{wandb_prefix}total_relevant_docs": total_relevant_docs,
        f"{wandb_prefix}avg_relevant_per_query": avg_relevant_per_query,
    }
    print(f"NDCG@1 {results[f'{wandb_prefix}ndcg_1']:.4f}  NDCG@5 {results[f'{wandb_prefix}ndcg_5']:.4f}  "
          f"NDCG@10 {results[f'{wandb_prefix}ndcg_10']:.4f} (±{results[f'{wandb_prefix}ndcg_10_std']:.4f})")
    if log_wandb:
        try:
            import wandb  # type: ignore

            wandb.log(results)
        except Exception:  # logging only
            pass
    return results
