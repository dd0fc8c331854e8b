"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not product code.

CPU restatement (torch-CPU, fp32 by default, fp64 on request) of the two-tower
triplet hot path of freemvmt/two-towers-overlords.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product package (`two-towers-overlords_amd/`) never does, and its
ops raise when the HIP library is missing.

Every function cites the reference lines it restates (paths relative to
/root/reference).  The restatement keeps the reference's op *sequence* (same
torch ops in the same order) so that (a) results agree with the reference to
fp32 rounding and (b) its wall time on host cores is a fair stand-in for the
reference CPU loop (the reference's Python cannot travel to the GPU box).

Pinning: the reference has no tests, golden vectors or fixtures of its own
(SURVEY.md §4), so this oracle is pinned by outputs of the reference itself,
generated in the build container by `tests/golden/make_golden.py` (which
imports /root/reference/backend/{model,training,data}.py and calls the
reference's own forward / TripletLoss / train_epoch / create_triplets) and
committed under `tests/golden/*.npz`.  `tests/test_oracle_golden.py` checks
this file against every one of them.

Inputs are token-id tensors, not strings: the tokenizer (backend/model.py:43-45)
is a third-party host component outside the hot path; its int64 outputs
(`input_ids`, `attention_mask`) are this path's inputs.
"""

from __future__ import annotations

import math
import random
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- #
# Tower forward
# --------------------------------------------------------------------------- #


def embed_lookup(table, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """backend/model.py:51-52, `output = self.pretrained_model(**tokens)`, `output[0]`.
    `embed` encoder mode (a tensor `table`): the pretrained model reduced to its word-embedding
    table, out[b,s,:] = table[input_ids[b,s],:].  `bert` mode (`table` is the frozen encoder
    module itself, e.g. a transformers BertModel): what the reference executes — the module is
    called with the tokenizer's outputs and its first output (last_hidden_state) is pooled."""
    if isinstance(table, torch.nn.Module):
        return table(input_ids=input_ids, attention_mask=attention_mask)[0]
    return F.embedding(input_ids, table)


def mean_pooling(token_embeddings: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """backend/model.py:63-72 (`_mean_pooling`), same op sequence:
    expand the mask over D, cast to float, sum(emb*mask, 1) / clamp(mask.sum(1), 1e-9)."""
    input_mask_expanded = (
        attention_mask.unsqueeze(-1).expand(token_embeddings.size()).to(token_embeddings.dtype)
    )
    return torch.sum(token_embeddings * input_mask_expanded, 1) / torch.clamp(
        input_mask_expanded.sum(1), min=1e-9
    )


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """backend/model.py:56 — F.normalize(p=2, dim=1): x / max(||x||_2, 1e-12)."""
    return F.normalize(x, p=2, dim=1)


def pool_normalize(token_embeddings: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """backend/model.py:55-56."""
    return l2_normalize(mean_pooling(token_embeddings, attention_mask))


def gather_pool_normalize(
    table: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor
) -> torch.Tensor:
    """backend/model.py:51-56 in `embed` mode: gather -> masked mean -> L2 normalise."""
    return pool_normalize(embed_lookup(table, input_ids), attention_mask)


def projection(
    x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor
) -> torch.Tensor:
    """backend/model.py:33-38,59 — Sequential(Linear(D,P), ReLU(), Linear(P,P))."""
    return F.linear(F.relu(F.linear(x, w1, b1)), w2, b2)


def tower_forward(
    table: torch.Tensor,
    input_ids: torch.Tensor,
    attention_mask: torch.Tensor,
    w1: torch.Tensor,
    b1: torch.Tensor,
    w2: torch.Tensor,
    b2: torch.Tensor,
) -> torch.Tensor:
    """backend/model.py:40-60 (`AveragePoolingTower.forward`) from token ids on."""
    with torch.no_grad():  # model.py:51 — the encoder is frozen and runs under no_grad
        token_embeddings = embed_lookup(table, input_ids, attention_mask)
    pooled = pool_normalize(token_embeddings, attention_mask)
    return projection(pooled, w1, b1, w2, b2)


# --------------------------------------------------------------------------- #
# Triplet loss
# --------------------------------------------------------------------------- #


def cosine_distance(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """backend/model.py:132-135 — 1 - F.cosine_similarity(x, y, dim=1) (eps 1e-8)."""
    return 1 - F.cosine_similarity(x, y, dim=1)


def triplet_loss(
    anchor: torch.Tensor, positive: torch.Tensor, negative: torch.Tensor, margin: float
) -> torch.Tensor:
    """backend/model.py:137-145 — mean(relu(pos_dist - neg_dist + margin))."""
    pos_dist = cosine_distance(anchor, positive)
    neg_dist = cosine_distance(anchor, negative)
    loss = F.relu(pos_dist - neg_dist + margin)
    return loss.mean()


def triplet_loss_closed_form(
    q: torch.Tensor, p: torch.Tensor, n: torch.Tensor, margin: float, grad_scale: Optional[float] = None
):
    """Closed-form forward + gradient of backend/model.py:137-145 (what
    `loss.backward()` at backend/training.py:50 produces for the three
    embeddings).  Independent derivation used to check both autograd and the
    HIP kernel.  With c = max(|a|, eps), u = a/c, v = b/max(|b|, eps):

        d cos(a,b)/da = (v - u * cos * r) / c,   r = c/|a| if |a| > 0 else 0

    r == 1 whenever |a| >= eps (the ordinary derivative).  For 0 < |a| < eps
    torch's cosine_similarity clamps the norm's *value* but differentiates the
    un-clamped norm, which is what r reproduces (pinned by
    tests/golden/loss_edges.npz rows 3, 5, 8).
    Returns (loss, pos_dist, neg_dist, dq, dp, dn)."""
    eps = 1e-8
    B = q.shape[0]
    g = (1.0 / B) if grad_scale is None else grad_scale

    def unit(x):
        nrm = x.norm(dim=1, keepdim=True)
        c = nrm.clamp_min(eps)
        r = torch.where(nrm > 0, c / nrm, torch.zeros_like(nrm))
        return x / c, c, r

    uq, cq, rq = unit(q)
    up, cp, rp = unit(p)
    un, cn, rn = unit(n)
    cos_p = (uq * up).sum(1)
    cos_n = (uq * un).sum(1)
    pos_dist = 1 - cos_p
    neg_dist = 1 - cos_n
    hinge = pos_dist - neg_dist + margin
    loss = F.relu(hinge).mean()
    act = (hinge > 0).to(q.dtype).unsqueeze(1) * g

    def dcos(ua, ca, ra, ub, cosab):
        return (ub - ua * (cosab.unsqueeze(1) * ra)) / ca

    dq = act * (-dcos(uq, cq, rq, up, cos_p) + dcos(uq, cq, rq, un, cos_n))
    dp = act * (-dcos(up, cp, rp, uq, cos_p))
    dn = act * (dcos(un, cn, rn, uq, cos_n))
    return loss, pos_dist, neg_dist, dq, dp, dn


# --------------------------------------------------------------------------- #
# Parameters, Adam, train step / epoch
# --------------------------------------------------------------------------- #

PARAM_KEYS = (
    "query_tower.projection.0.weight",
    "query_tower.projection.0.bias",
    "query_tower.projection.2.weight",
    "query_tower.projection.2.bias",
    "document_tower.projection.0.weight",
    "document_tower.projection.0.bias",
    "document_tower.projection.2.weight",
    "document_tower.projection.2.bias",
)
"""state_dict key names and `model.parameters()` order of the 8 trainable tensors
(backend/model.py:33-38,84-85; backend/search.py:104-106)."""


def init_params(D: int, P: int, seed: int, dtype=torch.float32) -> dict:
    """torch default Linear init, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight
    and bias (backend/model.py:35,37 leave the default), from a seeded generator."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for tower in ("query_tower", "document_tower"):
        for idx, (fo, fi) in ((0, (P, D)), (2, (P, P))):
            bound = 1.0 / math.sqrt(fi)
            out[f"{tower}.projection.{idx}.weight"] = (
                (torch.rand(fo, fi, generator=g, dtype=torch.float64) * 2 - 1) * bound
            ).to(dtype)
            out[f"{tower}.projection.{idx}.bias"] = (
                (torch.rand(fo, generator=g, dtype=torch.float64) * 2 - 1) * bound
            ).to(dtype)
    return out


def encode(table, ids, mask, params: dict, tower: str) -> torch.Tensor:
    """backend/model.py:87-93 (`encode_queries` / `encode_documents`)."""
    k = f"{tower}.projection."
    return tower_forward(
        table, ids, mask, params[k + "0.weight"], params[k + "0.bias"], params[k + "2.weight"], params[k + "2.bias"]
    )


@dataclass
class AdamState:
    """State of torch.optim.Adam(model.parameters(), lr) as built at
    backend/training.py:436 — betas (0.9, 0.999), eps 1e-8, weight_decay 0."""

    lr: float
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    step: int = 0
    m: Optional[dict] = None
    v: Optional[dict] = None


def adam_step(params: dict, grads: dict, st: AdamState) -> None:
    """One `optimizer.step()` (backend/training.py:51), restating the op order
    of torch's single-tensor Adam: m.lerp_(g, 1-b1); v = v*b2 + (1-b2) g*g;
    denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * m/denom."""
    if st.m is None:
        st.m = {k: torch.zeros_like(v) for k, v in params.items()}
        st.v = {k: torch.zeros_like(v) for k, v in params.items()}
    st.step += 1
    bc1 = 1 - st.beta1**st.step
    bc2 = 1 - st.beta2**st.step
    step_size = st.lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    for k, p in params.items():
        g = grads[k]
        st.m[k].lerp_(g, 1 - st.beta1)
        st.v[k].mul_(st.beta2).addcmul_(g, g, value=1 - st.beta2)
        denom = (st.v[k].sqrt() / bc2_sqrt).add_(st.eps)
        p.addcdiv_(st.m[k], denom, value=-step_size)


def loss_and_grads(table, batch, params: dict, margin: float, loss_scale: float = 1.0, doc_table=None):
    """backend/training.py:36-50 without the optimizer: three encodes in the
    reference's order (q, pos, neg), criterion, backward.  `batch` =
    ((ids_q, mask_q), (ids_p, mask_p), (ids_n, mask_n)).
    `doc_table`: the document tower's own copy of the (identical) frozen table — the reference
    loads one pretrained model PER tower (backend/model.py:84-85); values are the same, but two
    46.9 MB tables is what its CPU loop walks, so the timing leg passes a copy here.
    Returns (loss, grads dict, (q_emb, p_emb, n_emb))."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    (iq, mq), (ip, mp), (i_n, mn) = batch
    dt = table if doc_table is None else doc_table
    q = encode(table, iq, mq, leaves, "query_tower")
    p = encode(dt, ip, mp, leaves, "document_tower")
    n = encode(dt, i_n, mn, leaves, "document_tower")
    loss = triplet_loss(q, p, n, margin)
    (loss * loss_scale).backward()
    grads = {k: v.grad.detach() for k, v in leaves.items()}
    return loss.detach(), grads, (q.detach(), p.detach(), n.detach())


def train_epoch(table, batches: Iterable, params: dict, st: AdamState, margin: float,
                doc_table=None) -> Tuple[float, List[float]]:
    """backend/training.py:26-63 (`train_epoch`): per batch zero_grad -> 3 encodes
    -> loss -> backward -> Adam step; returns mean of per-batch losses (and the trace).
    `params` is updated in place."""
    total, trace = 0.0, []
    for batch in batches:
        loss, grads, _ = loss_and_grads(table, batch, params, margin, doc_table=doc_table)
        adam_step(params, grads, st)
        total += loss.item()
        trace.append(loss.item())
    return total / len(trace), trace


def train_epoch_accum(
    table, batches: Sequence, params: dict, st: AdamState, margin: float, accumulation_steps: int,
    carry: Optional[dict] = None,
) -> Tuple[float, List[float], dict]:
    """fp32 restatement of the accumulation semantics of backend/training.py:80-104
    (`train_epoch_optimized` minus autocast/GradScaler): loss/k per batch,
    optimizer step + zero_grad every k-th batch; left-over grads at the end of the
    epoch are neither applied nor cleared (returned as `carry`, so they leak into
    the next epoch exactly as in the reference)."""
    acc = carry
    total, trace = 0.0, []
    for batch_idx, batch in enumerate(batches):
        loss, grads, _ = loss_and_grads(table, batch, params, margin, loss_scale=1.0 / accumulation_steps)
        if acc is None:
            acc = {k: g.clone() for k, g in grads.items()}
        else:
            for k in acc:
                acc[k] += grads[k]
        if (batch_idx + 1) % accumulation_steps == 0:
            adam_step(params, acc, st)
            acc = None
        total += loss.item()
        trace.append(loss.item())
    return total / len(trace), trace, acc


# --------------------------------------------------------------------------- #
# In-batch negative sampling
# --------------------------------------------------------------------------- #


def create_triplet_indices(query_ids: Sequence[int], rng: random.Random,
                           positive_nonempty: Optional[Sequence[bool]] = None) -> List[int]:
    """backend/data.py:113-137 (`create_triplets` + `_get_random_negative`) on
    indices: for item i draw j uniformly from the other in-batch items
    (`rng.choice([j for j != i])`, same call sequence as the reference so a
    shared seed gives the same draws), reject while query_id equal OR item j's
    positive text is empty (data.py:126 `if negative and ...`: an empty string is
    falsy); the negative *text* is item j's positive.  Returns neg_idx[i] = j."""
    n = len(query_ids)
    out = []
    for i in range(n):
        while True:
            j = rng.choice([k for k in range(n) if k != i])
            if (positive_nonempty is None or positive_nonempty[j]) and query_ids[j] != query_ids[i]:
                out.append(j)
                break
    return out


# --------------------------------------------------------------------------- #
# Closed-form KAT-0 inputs — RNG-free, reproducible anywhere
# --------------------------------------------------------------------------- #


def kat0_inputs():
    """KAT-0: V=64, D=8, S=6, B=4, P=4; everything closed-form (computed in f64,
    cast to f32).  Returns (table, batch, params, margin)."""
    V, D, S, B, P = 64, 8, 6, 4, 4
    v = torch.arange(V, dtype=torch.float64).unsqueeze(1)
    d = torch.arange(D, dtype=torch.float64).unsqueeze(0)
    table = torch.sin(0.37 * v + 0.11 * d).float()
    lens = torch.tensor([6, 4, 1, 3])
    s = torch.arange(S).unsqueeze(0)
    mask = (s < lens.unsqueeze(1)).long()
    b = torch.arange(B).unsqueeze(1)
    batch = []
    for role in range(3):
        ids = ((7 * b + 3 * s + 11 * role) % 64) * mask
        batch.append((ids.long(), mask.clone()))
    params = {}
    for tower, phi in (("query_tower", 0.0), ("document_tower", 0.5)):
        pp = torch.arange(P, dtype=torch.float64).unsqueeze(1)
        dd = torch.arange(D, dtype=torch.float64).unsqueeze(0)
        qq = torch.arange(P, dtype=torch.float64).unsqueeze(0)
        params[f"{tower}.projection.0.weight"] = (torch.cos(0.13 * pp + 0.07 * dd + phi) / 4).float()
        params[f"{tower}.projection.0.bias"] = (0.01 * pp.squeeze(1)).float()
        params[f"{tower}.projection.2.weight"] = (torch.sin(0.05 * pp - 0.09 * qq + phi) / 2).float()
        params[f"{tower}.projection.2.bias"] = (-0.02 * pp.squeeze(1)).float()
    return table, tuple(batch), params, 0.3


# --------------------------------------------------------------------------- #
# Seeded synthetic workloads (shared by tests and bench's cpu_baseline leg)
# --------------------------------------------------------------------------- #


def synthetic_table(V: int, D: int, seed: int = 1) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(V, D, generator=g) * 0.02


def synthetic_batch(B: int, S: int, V: int, seed: int, ragged: bool = False, dtype=torch.int64):
    """Three (ids, mask) pairs.  ragged: len ~ U{8..S} for queries, U{min(32,S)..S} for docs."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for role in range(3):
        ids = torch.randint(0, V, (B, S), generator=g, dtype=torch.int64)
        if ragged:
            lo = min(8, S) if role == 0 else min(32, S)
            lens = torch.randint(lo, S + 1, (B,), generator=g)
        else:
            lens = torch.full((B,), S)
        mask = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).to(torch.int64)
        out.append(((ids * mask).to(dtype), mask))
    return tuple(out)


# --------------------------------------------------------------------------- #
# Evaluation arithmetic (backend/training.py:281-311)
# --------------------------------------------------------------------------- #


def cosine_scores(q: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """Row i = torch.cosine_similarity(q[i:i+1], d, dim=1) — backend/training.py:297-299 for every query."""
    return torch.stack([F.cosine_similarity(q[i: i + 1], d, dim=1) for i in range(q.shape[0])])


def ndcg_rows(scores, gains, k: int, valid=None):
    """sklearn.metrics.ndcg_score(true_relevance.reshape(1,-1), similarities.reshape(1,-1), k=k) per row
    (backend/training.py:301-311); `valid` restricts a row to its candidate pool.  sklearn is the
    third-party implementation the reference calls (scikit-learn, pinned in backend/uv.lock)."""
    import numpy as np
    from sklearn.metrics import ndcg_score

    scores, gains = np.asarray(scores, dtype=np.float64), np.asarray(gains, dtype=np.float64)
    out = []
    for i in range(scores.shape[0]):
        sel = slice(None) if valid is None else np.asarray(valid[i]).astype(bool)
        s, g = scores[i][sel], gains[i][sel]
        if len(s) < 2:  # sklearn refuses a single document; the score is 1 if it is relevant else 0
            out.append(float(g.sum() > 0))
            continue
        out.append(float(ndcg_score(g.reshape(1, -1), s.reshape(1, -1), k=k)))
    return np.array(out)


# --------------------------------------------------------------------------- #
# In-batch-negatives triplet loss — NO reference implementation: this is the
# definition the HIP path is checked against ("parity unpinned").
# --------------------------------------------------------------------------- #


def inbatch_triplet_loss(q, d, margin, pos_index=None, qid_q=None, qid_d=None, denominator=None):
    """sum_i sum_{j valid} relu(cos(q_i,d_j) - cos(q_i,d_pos_i) + margin) / denominator; valid j: j != pos_i
    and (no ids or qid_d[j] != qid_q[i]).  Default denominator Bq*(Bd-1)."""
    Bq, Bd = q.shape[0], d.shape[0]
    pos = torch.arange(Bq) if pos_index is None else pos_index.long()
    S = torch.stack([F.cosine_similarity(q[i: i + 1], d, dim=1) for i in range(Bq)])
    s_pos = S[torch.arange(Bq), pos].unsqueeze(1)
    valid = torch.ones(Bq, Bd, dtype=torch.bool)
    valid[torch.arange(Bq), pos] = False
    if qid_q is not None:
        valid &= qid_d.unsqueeze(0) != qid_q.unsqueeze(1)
    h = F.relu(S - s_pos + margin) * valid
    return h.sum() / (float(Bq * (Bd - 1)) if denominator is None else denominator)
