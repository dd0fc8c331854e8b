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

import random
from typing import Union

import numpy as np
import torch

from . import ops
from .model import TwoTowersModel


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
) -> Union[float, dict]:
    """NDCG@{10,5,1} over sampled query groups.  Mirrors backend/training.py:136-380."""
    print(f"Sampling documents for evaluation (initial sample size: {sample_size})...")
    model.eval()

    # -- group by query_id until enough unique queries (training.py:184-201): identical `random` use
    query_groups: dict = {}
    sample_multiplier = 1
    while len(query_groups) < min_query_groups and sample_multiplier <= 10:
        sample_size_current = min(sample_size * sample_multiplier, len(dataset))
        indices = random.sample(range(len(dataset)), sample_size_current)
        for item in (dataset[i] for i in indices):
            query_id = item["query_id"]
            if query_id not in query_groups:
                query_groups[query_id] = {"query": item["query"], "relevant_docs": set()}
            query_groups[query_id]["relevant_docs"].add(item["positive"])
        sample_multiplier += 1

    query_ids = sorted(query_groups.keys(), key=lambda qid: len(query_groups[qid]["relevant_docs"]), reverse=True)[
        :min_query_groups
    ]
    total_relevant_docs = sum(len(query_groups[qid]["relevant_docs"]) for qid in query_ids)
    avg_relevant_per_query = total_relevant_docs / len(query_ids)
    print(f"Selected {len(query_ids)} queries for evaluation")

    # -- candidate pools (training.py:230-277) as (unique document list, per-query membership)
    if candidate_pool_size == -1:
        docs = list(dataset.get_unique_passages())
        pools = None
    else:
        pools = []
        for query_id in query_ids:
            relevant_docs = query_groups[query_id]["relevant_docs"]
            irrelevant_docs = set(
                doc
                for other_qid in query_ids
                for doc in query_groups[other_qid]["relevant_docs"]
                if other_qid != query_id
                if doc not in relevant_docs
            )
            if len(irrelevant_docs) > candidate_pool_size:
                irrelevant_docs = random.sample(list(irrelevant_docs), candidate_pool_size)
            pools.append(list(relevant_docs) + list(irrelevant_docs))
        docs = list(dict.fromkeys(d for pool in pools for d in pool))
    doc_index = {d: j for j, d in enumerate(docs)}
    Q, N = len(query_ids), len(docs)

    relevance = np.zeros((Q, N), dtype=np.float32)
    valid = None if pools is None else np.zeros((Q, N), dtype=np.uint8)
    for i, query_id in enumerate(query_ids):
        for d in query_groups[query_id]["relevant_docs"]:
            j = doc_index.get(d)
            if j is not None:
                relevance[i, j] = 1.0
        if pools is not None:
            valid[i, [doc_index[d] for d in pools[i]]] = 1

    # -- embeddings through the hot path, similarities and NDCG on the device
    with torch.no_grad():
        query_embeds = model.encode_queries([query_groups[qid]["query"] for qid in query_ids])
        doc_embeds = model.encode_documents_batched(docs, batch_size=batch_size)
        scores = ops.cosine_scores(query_embeds.contiguous(), doc_embeds.contiguous())
        dev = scores.device
        ndcg = ops.ndcg_at_k(
            scores,
            torch.from_numpy(relevance).to(dev),
            ks=(10, 5, 1),
            valid=None if valid is None else torch.from_numpy(valid).to(dev),
        ).double().cpu().numpy()

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
