"""The input boundary with a REAL tokenizer: `transformers.BertTokenizerFast` built offline from
a vocab file this test writes, called exactly like backend/model.py:43-45 (`tokenizer(texts, padding=True,
truncation=True, return_tensors="pt", max_length=512)`).  Its int64 `input_ids` / `attention_mask` must reach the
gather bit for bit, batches must be padded to the longest text of the call like `padding=True` pads them, and the
string-level entry points (`TwoTowersModel.forward`, `TokenTripletLoader`, `run_training`) must agree with the oracle
on those very ids."""
import os
import random
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import two_towers_oracle as orc  # noqa: E402

pytestmark = pytest.mark.gpu

WORDS = ("what is the capital of france paris river long a an and run walk city country where who when how many people "
         "live in on at by for with from to europe asia africa america ocean sea lake mountain tall deep old new first last "
         "big small red blue green king queen president war peace year day night food water bread rice fish").split()
PIECES = ["##s", "##ing", "##ed", "##er", "##ly", "##est", ".", "?", ",", "!"]


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    transformers = pytest.importorskip("transformers")
    d = tmp_path_factory.mktemp("vocab")
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"[unused{i}]" for i in range(20)] + WORDS + PIECES
    (d / "vocab.txt").write_text("\n".join(vocab) + "\n")
    t = transformers.BertTokenizerFast.from_pretrained(str(d))  # a local directory: nothing is fetched
    assert t.is_fast and t.vocab_size == len(vocab)
    return t


def _texts(n, rng, lo=1, hi=14):
    out = []
    for _ in range(n):
        ws = [rng.choice(WORDS) + rng.choice(["", "", "s", "ing", "ed"]) for _ in range(rng.randint(lo, hi))]
        out.append(" ".join(ws).capitalize() + rng.choice(["?", ".", ""]))
    return out


def _model(tok, P, seed):
    from two_towers_overlords_amd.model import EmbeddingOnlyEncoder, TwoTowersModel

    table = orc.synthetic_table(tok.vocab_size, 64, seed=seed) * 30
    torch.manual_seed(seed)
    return TwoTowersModel(P, tokenizer=tok, pretrained_model=EmbeddingOnlyEncoder(table)).cuda(), table


def test_real_tokenizer_ids_reach_the_gather_bit_exactly_and_forward_matches_the_oracle(tok):
    from two_towers_overlords_amd import ops

    rng = random.Random(1)
    queries, docs = _texts(37, rng, 1, 8), _texts(37, rng, 3, 40)
    model, table = _model(tok, 48, 3)
    for texts, tower, name in ((queries, model.query_tower, "query_tower"), (docs, model.document_tower, "document_tower")):
        enc = tok(texts, padding=True, truncation=True, return_tensors="pt", max_length=512)  # backend/model.py:43-45
        ids, mask = enc["input_ids"], enc["attention_mask"]
        assert ids.dtype == torch.int64 and mask.dtype == torch.int64
        assert ids.shape[1] == int(mask.sum(1).max())  # padding=True: the longest text of THIS call, no more
        assert (ids[:, 0] == tok.cls_token_id).all() and ((ids == tok.sep_token_id).sum(1) == 1).all()  # [CLS] .. [SEP] are pooled too
        echo = ops.embed_gather(table.cuda(), ids.cuda())           # the rows K1 reads for these int64 ids ...
        assert torch.equal(echo.cpu(), table[ids])                  # ... are the reference's gather, bit for bit
        got = tower(texts)                                          # strings -> tokenizer -> .to(device) -> K1 -> K2
        p = {k: v.detach().cpu() for k, v in tower.projection.state_dict().items()}
        ref = orc.tower_forward(table, ids, mask, p["0.weight"], p["0.bias"], p["2.weight"], p["2.bias"])
        assert (got.cpu() - ref).abs().max().item() <= 1e-5, name
    q, d = model(queries, docs)
    assert torch.equal(q, model.encode_queries(queries)) and torch.equal(d, model.encode_documents_batched(docs, batch_size=16))


def test_real_tokenizer_through_the_token_loader_packed_files_and_run_training(tok, tmp_path):
    from two_towers_overlords_amd.data import TokenTripletLoader
    from two_towers_overlords_amd.model import EmbeddingOnlyEncoder
    from two_towers_overlords_amd.training import run_training

    rng = random.Random(2)
    n = 96
    qtext = _texts(24, rng, 1, 8)
    items = [{"query_id": i % 24, "query": qtext[i % 24], "positive": t} for i, t in enumerate(_texts(n, rng, 3, 30))]
    ld = TokenTripletLoader(items, tok, batch_size=32, device="cuda", seed=4, sampler="reference", emit="ids3")
    # tokenised once, rows = the tokenizer's own ids (int32 copies of its int64 output), lengths = its mask sums
    for texts, ids_t, lens_t in (([it["query"] for it in items], ld.q_ids, ld.q_lens), ([it["positive"] for it in items], ld.p_ids, ld.p_lens)):
        enc = tok(texts, padding=True, truncation=True, return_tensors="pt", max_length=512)
        assert torch.equal(ids_t.cpu().long(), enc["input_ids"]) and torch.equal(lens_t.cpu().long(), enc["attention_mask"].sum(1))
    for ids, lens in ld:  # pad="batch": every batch is cut to its own longest row, like padding=True per call
        assert ids.shape[2] == int(lens.max()) and ids.dtype == torch.int32
    # PackedTokens round trip (f3): ids.bin + offsets.bin and back
    prefix = str(tmp_path / "marco")
    ld.save_packed(prefix)
    ld2 = TokenTripletLoader.from_packed(prefix, batch_size=32, device="cuda", seed=4, sampler="reference", emit="ids3")
    assert torch.equal(ld2.q_ids, ld.q_ids) and torch.equal(ld2.p_ids, ld.p_ids) and torch.equal(ld2.p_lens, ld.p_lens)
    assert ld2.query_ids == ld.query_ids
    ld3 = TokenTripletLoader(items, tok, batch_size=32, device="cuda", seed=4, sampler="reference", emit="ids3")
    for a, b in zip(ld3, ld2):  # same seed, same draws (both loaders fresh)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(items)

        def __getitem__(self, i):
            return items[i]

    table = orc.synthetic_table(tok.vocab_size, 64, seed=5) * 30
    flat = {}
    for cache in (True, False):
        torch.manual_seed(0)
        m = run_training(num_epochs=2, batch_size=32, learning_rate=1e-2, max_samples=10 ** 6, projection_dim=32, margin=0.3,
                         use_wandb=False, accumulation_steps=1, train_dataset=DS(), tokenizer=tok,
                         pretrained_model=EmbeddingOnlyEncoder(table), seed=5, feature_cache=cache)
        flat[cache] = torch.cat([p.detach().reshape(-1).cpu() for p in m.trainable_parameters()])
        assert torch.isfinite(flat[cache]).all()
    assert (flat[True] - flat[False]).abs().max() <= 2e-5
