"""Host-side mirror of the reference's backend/model.py (same class names,
constructor arguments, methods, attribute names and state_dict keys), with the
arithmetic done by the HIP kernels behind the C ABI.

    reference                              here
    -------------------------------------  ------------------------------------------
    AveragePoolingTower  (model.py:13-72)   AveragePoolingTower  (+ encode_ids, encoder_mode)
    TwoTowersModel       (model.py:75-121)  TwoTowersModel       (+ encode_*_ids)
    TripletLoss          (model.py:125-145) TripletLoss

Additive only: tensor-level entry points (`encode_ids`) that take the
tokenizer's int64 `input_ids` / `attention_mask` directly, and
`encoder_mode={"embed","bert"}`: "embed" pools the frozen
word-embedding rows (the gather hot path), "bert" runs the frozen encoder with
stock PyTorch-ROCm ops and pools its last_hidden_state (reference-exact).  Left
unset, a `pretrained_model` that has transformer layers runs in "bert" mode — what
backend/model.py:51-52 executes, so reference-trained projections give the
reference's embeddings — and says so once; a bare embedding table runs "embed".
"""
from __future__ import annotations

import warnings
from typing import Callable, Optional, Tuple

import torch
from torch import Tensor, nn

from . import ops

DEFAULT_MODEL_NAME = "sentence-transformers/all-MiniLM-L6-v2"


class _ProjectionFn(torch.autograd.Function):
    """Linear(D,P) -> ReLU -> Linear(P,P) (backend/model.py:33-38,59) on the fp32
    MFMA kernels; gradients for the four parameters only (the input comes from the
    frozen encoder under no_grad, backend/model.py:29-30,51)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x = x.contiguous()
        out, h = ops.proj_fwd(x, w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous())
        ctx.save_for_backward(x, h, w2)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, h, w2 = ctx.saved_tensors
        dW1, db1, dW2, db2 = ops.proj_bwd(d_out.contiguous(), x, h, w2.contiguous())
        return None, dW1, db1, dW2, db2


class _TripletLossFn(torch.autograd.Function):
    """mean(relu((1-cos(a,p)) - (1-cos(a,n)) + margin)) (backend/model.py:137-145);
    the kernel produces the loss and d loss/d{a,p,n} in one pass."""

    @staticmethod
    def forward(ctx, anchor, positive, negative, margin):
        loss, _, _, dq, dp, dn = ops.triplet_fwd_bwd(
            anchor.contiguous(), positive.contiguous(), negative.contiguous(), float(margin)
        )
        ctx.save_for_backward(dq._base)  # dq, dp, dn are the three rows of ONE [3, B, P] buffer
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        gd = g * d  # one launch for the three gradients
        return gd[0], gd[1], gd[2], None


class _CosineDistanceFn(torch.autograd.Function):
    """1 - F.cosine_similarity(x, y, dim=1) (backend/model.py:132-135) with torch's gradient, on the device."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = x.contiguous(), y.contiguous()
        ctx.save_for_backward(x, y)
        return ops.cosine_distance(x, y)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        dx, dy = ops.cosine_distance_bwd(x, y, g.contiguous())
        return dx, dy


class AveragePoolingTower(nn.Module):
    """Frozen token embeddings -> masked average pool -> L2 normalise -> trainable
    projection.  Mirrors backend/model.py:13-72.

    `tokenizer` / `pretrained_model` may be injected (offline use, tests, sharing
    one frozen model between towers); otherwise they are loaded exactly like the
    reference does (`AutoTokenizer/AutoModel.from_pretrained(model_name)`,
    model.py:23-24), which needs the files locally or network access.
    """

    def __init__(
        self,
        model_name: str = DEFAULT_MODEL_NAME,
        projection_dim: int = 128,
        *,
        tokenizer: Optional[Callable] = None,
        pretrained_model: Optional[nn.Module] = None,
        encoder_mode: Optional[str] = None,
    ):
        super().__init__()
        if encoder_mode not in (None, "embed", "bert"):
            raise ValueError("encoder_mode must be 'embed' or 'bert'")
        if tokenizer is None or pretrained_model is None:
            from transformers import AutoModel, AutoTokenizer  # same loaders as model.py:23-24

            tokenizer = tokenizer or AutoTokenizer.from_pretrained(model_name)
            pretrained_model = pretrained_model or AutoModel.from_pretrained(model_name)
        if encoder_mode is None:
            encoder_mode = _default_encoder_mode(pretrained_model)
        self.tokenizer = tokenizer
        self.pretrained_model = pretrained_model
        self.encoder_mode = encoder_mode
        self.embedding_dim = int(_hidden_size(pretrained_model))
        for param in self.pretrained_model.parameters():  # model.py:29-30
            param.requires_grad = False
        # same container and default init as model.py:33-38, so state_dict keys are
        # projection.0.{weight,bias} / projection.2.{weight,bias}
        self.projection = nn.Sequential(
            nn.Linear(self.embedding_dim, projection_dim),
            nn.ReLU(),
            nn.Linear(projection_dim, projection_dim),
        )

    # -- the encoder mode travels with the weights -------------------------------------
    # as state_dict METADATA (the per-module dict torch keeps beside `version`), not as a key: the reference loads
    # these checkpoints with a strict load_state_dict (search.py:229-231) and must not see keys it does not know.
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        meta = getattr(destination, "_metadata", None)
        if meta is not None and prefix[:-1] in meta:
            meta[prefix[:-1]]["encoder_mode"] = self.encoder_mode

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        trained = (local_metadata or {}).get("encoder_mode")
        if trained is None and getattr(self.pretrained_model, "encoder", None) is not None:
            trained = "bert"  # no record: a reference checkpoint, trained behind the full frozen encoder (model.py:51-52)
        if trained is not None and trained != self.encoder_mode and any(k.startswith(prefix + "projection.") for k in state_dict):
            warnings.warn(f"{prefix or 'tower.'}projection weights were trained with encoder_mode='{trained}' but this tower "
                          f"runs encoder_mode='{self.encoder_mode}': its embeddings will not match the training run's; "
                          f"construct the model with encoder_mode='{trained}'.", stacklevel=2)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    # -- the reference entry point ------------------------------------------------
    def forward(self, texts: list[str]) -> Tensor:
        """model.py:40-60: tokenize, move to the model's device, pool, project.  `embed` mode behind a transformers
        tokenizer: the same ids reach the gather as int32 rows with TT_PAD_ID behind each text's tokens
        (textstore.hf_token_rows) — a quarter of the bytes of int64 ids + int64 mask over PCIe, and none of transformers'
        Python-side tensor conversion, which costs more than the tokenisation."""
        if self.encoder_mode == "embed" and len(texts):
            from .textstore import hf_token_rows, is_hf_tokenizer

            if is_hf_tokenizer(self.tokenizer):
                rows, _ = hf_token_rows(self.tokenizer, texts)
                ids = torch.from_numpy(rows).to(_module_device(self.pretrained_model))
                with torch.no_grad():
                    pooled = ops.gather_pool_norm(self.embedding_table().detach(), ids)
                return self.project(pooled)
        return self.encode_tokens(self.tokenize(texts))

    def tokenize(self, texts: list[str]) -> dict:
        """model.py:43-48: the tokenizer call and the `.to(device)` of its tensors.  A right-padding transformers tokenizer is
        asked for plain lists and its padded int64 `input_ids` / `attention_mask` (/ all-zero `token_type_ids`) are rebuilt
        here — the tensors the reference's call returns, element for element (tests/test_textstore_cpu.py), without
        transformers' Python-side tensor conversion (textstore.hf_token_rows)."""
        device = _module_device(self.pretrained_model)
        tok = self.tokenizer
        if len(texts) and getattr(tok, "padding_side", None) == "right":
            from .textstore import PAD32, hf_token_rows, is_hf_tokenizer

            if is_hf_tokenizer(tok):
                rows, _ = hf_token_rows(tok, texts)
                rows = torch.from_numpy(rows).to(device)
                keep = rows != PAD32
                pad_id = tok.pad_token_id if tok.pad_token_id is not None else 0
                out = {"input_ids": torch.where(keep, rows, torch.full((), pad_id, dtype=torch.int32, device=device)).long()}
                if "token_type_ids" in getattr(tok, "model_input_names", ()):
                    out["token_type_ids"] = torch.zeros_like(out["input_ids"])
                out["attention_mask"] = keep.long()
                return out
        tokens = tok(texts, padding=True, truncation=True, return_tensors="pt", max_length=512)
        return {k: v.to(device) for k, v in tokens.items()}

    # -- tensor-level entry points (additive) -------------------------------------
    def embedding_table(self) -> Tensor:
        """The frozen word-embedding matrix [V, D] the `embed` mode gathers from."""
        pm = self.pretrained_model
        if hasattr(pm, "get_input_embeddings"):
            return pm.get_input_embeddings().weight
        return pm.word_embeddings.weight

    def pool_tokens(self, tokens: dict) -> Tensor:
        """model.py:51-56 -> pooled, normalised [B, D] (no grad)."""
        ids, mask = tokens["input_ids"], tokens.get("attention_mask")
        with torch.no_grad():
            if self.encoder_mode == "embed":
                return ops.gather_pool_norm(self.embedding_table().detach(), ids.contiguous(),
                                            None if mask is None else mask.contiguous())
            output = self.pretrained_model(**tokens)  # frozen encoder, stock PyTorch-ROCm ops
            hidden = output[0].float().contiguous()
            return ops.pool_norm_dense(hidden, None if mask is None else mask.contiguous())

    def encode_tokens(self, tokens: dict) -> Tensor:
        return self.project(self.pool_tokens(tokens))

    def project(self, pooled: Tensor) -> Tensor:
        """model.py:59: the trainable Linear -> ReLU -> Linear on pooled, normalised rows [B, D]."""
        p = self.projection
        return _ProjectionFn.apply(pooled, p[0].weight, p[0].bias, p[2].weight, p[2].bias)

    def encode_ids(self, input_ids: Tensor, attention_mask: Optional[Tensor] = None) -> Tensor:
        tokens = {"input_ids": input_ids}
        if attention_mask is not None:
            tokens["attention_mask"] = attention_mask
        return self.encode_tokens(tokens)


class TwoTowersModel(nn.Module):
    """Two-towers architecture for document retrieval.  Mirrors backend/model.py:75-121.

    The reference builds two private copies of the same frozen MiniLM
    (model.py:84-85); here both towers may share one `pretrained_model` object —
    `state_dict()` still lists it under both `query_tower.` and `document_tower.`
    prefixes, so checkpoints interchange with the reference (search.py:104-106,229-231).
    """

    def __init__(
        self,
        projection_dim: int = 128,
        model_name: str = DEFAULT_MODEL_NAME,
        *,
        tokenizer: Optional[Callable] = None,
        pretrained_model: Optional[nn.Module] = None,
        encoder_mode: Optional[str] = None,
    ):
        super().__init__()
        kw = dict(tokenizer=tokenizer, pretrained_model=pretrained_model, encoder_mode=encoder_mode)
        self.query_tower = AveragePoolingTower(model_name, projection_dim, **kw)
        if pretrained_model is None:  # loaded from disk: share the frozen weights and tokenizer
            kw = dict(tokenizer=self.query_tower.tokenizer, pretrained_model=self.query_tower.pretrained_model,
                      encoder_mode=self.query_tower.encoder_mode)
        self.document_tower = AveragePoolingTower(model_name, projection_dim, **kw)

    def encode_queries(self, queries: list[str]) -> Tensor:
        return self.query_tower(queries)

    def encode_documents(self, documents: list[str]) -> Tensor:
        return self.document_tower(documents)

    def encode_documents_batched(self, documents: list[str], batch_size: int = 1024) -> Tensor:
        """model.py:95-112 without the per-chunk `.cpu()` round trip: chunks are
        encoded under the same batch-size control and concatenated on the device."""
        if len(documents) <= batch_size:
            return self.encode_documents(documents)
        embeddings = []
        for i in range(0, len(documents), batch_size):
            embeddings.append(self.encode_documents(documents[i: i + batch_size]))
        return torch.cat(embeddings, dim=0)

    def forward(self, queries: list[str], documents: list[str]) -> Tuple[Tensor, Tensor]:
        return self.encode_queries(queries), self.encode_documents(documents)

    # tensor-level (additive)
    def encode_query_ids(self, input_ids: Tensor, attention_mask: Optional[Tensor] = None) -> Tensor:
        return self.query_tower.encode_ids(input_ids, attention_mask)

    def encode_document_ids(self, input_ids: Tensor, attention_mask: Optional[Tensor] = None) -> Tensor:
        return self.document_tower.encode_ids(input_ids, attention_mask)

    def trainable_parameters(self):
        """The 8 trainable tensors in `model.parameters()` order (model.py:33-38,84-85)."""
        out = []
        for tower in (self.query_tower, self.document_tower):
            p = tower.projection
            out += [p[0].weight, p[0].bias, p[2].weight, p[2].bias]
        return out


class TripletLoss(nn.Module):
    """Triplet loss with cosine distance.  Mirrors backend/model.py:125-145."""

    def __init__(self, margin: float = 0.3):
        super().__init__()
        self.margin = margin

    def cosine_distance(self, x: Tensor, y: Tensor) -> Tensor:
        """1 - cosine_similarity(x, y, dim=1) (model.py:132-135); differentiable like the reference's plain-torch
        method, so a caller may build its own loss from it (`forward` fuses the whole triplet loss instead)."""
        return _CosineDistanceFn.apply(x, y)

    def forward(self, anchor: Tensor, positive: Tensor, negative: Tensor) -> Tensor:
        return _TripletLossFn.apply(anchor, positive, negative, self.margin)


class _InBatchLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, documents, margin, query_ids):
        loss, dq, dd = ops.inbatch_triplet_fwd_bwd(anchor.contiguous(), documents.contiguous(), float(margin),
                                                   qid_q=query_ids, qid_d=query_ids)
        ctx.save_for_backward(dq, dd)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dq, dd = ctx.saved_tensors
        return g * dq, g * dd, None, None


class InBatchTripletLoss(nn.Module):
    """Triplet loss over ALL in-batch negatives (BASELINE.json configs[4]; not in the reference, whose
    `TripletDataLoader` samples one in-batch positive per item as the negative): for query i, document i
    is the positive and every other document j of the batch a negative,
        mean_{i, j != i} relu((1 - cos(q_i, d_i)) - (1 - cos(q_i, d_j)) + margin);
    with `query_ids`, documents of the same query are not used as negatives (like data.py:124-128)."""

    def __init__(self, margin: float = 0.3):
        super().__init__()
        self.margin = margin

    def forward(self, anchor: Tensor, documents: Tensor, query_ids: Optional[Tensor] = None) -> Tensor:
        qid = None if query_ids is None else query_ids.to(torch.int64).contiguous()
        return _InBatchLossFn.apply(anchor, documents, self.margin, qid)


class EmbeddingOnlyEncoder(nn.Module):
    """A frozen `pretrained_model` that is just a word-embedding table — what the
    `embed` encoder mode needs, constructible offline (synthetic or loaded rows).
    `forward(input_ids=...)` returns `(embeddings,)` like an HF model's first output."""

    def __init__(self, table: Tensor):
        super().__init__()
        self.word_embeddings = nn.Embedding.from_pretrained(table, freeze=True)

        class _Cfg:
            hidden_size = table.shape[1]
            vocab_size = table.shape[0]

        self.config = _Cfg()

    def get_input_embeddings(self):
        return self.word_embeddings

    @property
    def device(self):
        return self.word_embeddings.weight.device

    def forward(self, input_ids=None, **kw):
        return (self.word_embeddings(input_ids),)


_warned_default_bert = False


def _default_encoder_mode(pm: nn.Module) -> str:
    """encoder_mode left unset: a model with transformer layers (HF `BertModel.encoder`) runs as the
    reference's code does (backend/model.py:51-52, the full frozen encoder); a bare embedding table
    has nothing but the `embed` path.  Never silently the faster of the two."""
    global _warned_default_bert
    if getattr(pm, "encoder", None) is None:
        return "embed"
    if not _warned_default_bert:
        _warned_default_bert = True
        warnings.warn("encoder_mode not given and the pretrained model has encoder layers: running the frozen "
                      "encoder (encoder_mode='bert', what backend/model.py:51-52 executes, stock PyTorch ops).  "
                      "Pass encoder_mode='embed' to pool the word-embedding rows directly (the HIP gather hot "
                      "path, what the reference's docstrings describe); projections trained in one mode do not "
                      "transfer to the other.", stacklevel=3)
    return "bert"


def _hidden_size(pm: nn.Module) -> int:
    cfg = getattr(pm, "config", None)
    if cfg is not None and hasattr(cfg, "hidden_size"):
        return cfg.hidden_size  # model.py:26
    return pm.get_input_embeddings().weight.shape[1]


def _module_device(pm: nn.Module):
    dev = getattr(pm, "device", None)  # model.py:48 uses pretrained_model.device
    if dev is not None:
        return dev
    return next(pm.parameters()).device
