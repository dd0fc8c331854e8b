"""Pins oracle/two_towers_oracle.py against outputs of the reference itself
(tests/golden/*.npz, made by tests/golden/make_golden.py in the build container)."""
import random

import numpy as np
import pytest
import torch

from oracle import two_towers_oracle as orc
from conftest import load_golden, assert_params_close

T = torch.from_numpy
TOL = 1e-6  # oracle-vs-reference budget (same torch ops, same order; threads may differ)


def _batches(g):
    nb = int(g["n_batches"])
    return [
        tuple((T(g[f"batch{i}/ids_{r}"]), T(g[f"batch{i}/mask_{r}"])) for r in "qpn")
        for i in range(nb)
    ]


def _params(g, prefix):
    return {k: T(g[prefix + k]).clone() for k in orc.PARAM_KEYS}


def test_kat0_survey_values():
    """KAT-0 known answers recorded in SURVEY.md §4 (measured through the reference)."""
    table, batch, params, margin = orc.kat0_inputs()
    loss, grads, (q, p, n) = orc.loss_and_grads(table, batch, params, margin)
    assert abs(loss.item() - 0.551760912) < 1e-7
    np.testing.assert_allclose(orc.cosine_distance(q, p).numpy(), [5.67436218e-04, 1.83597541, 0, 1.95011449], atol=2e-7)
    np.testing.assert_allclose(orc.cosine_distance(q, n).numpy(), [1.85364747, 1.79620230, 1.83395767, 0.38284397], atol=2e-7)
    np.testing.assert_allclose(q[0].numpy(), [-0.00207082, -0.02081532, -0.03955777, -0.05830134], atol=1e-7)
    assert abs(grads["query_tower.projection.0.weight"].norm().item() - 0.160503060) < 1e-7
    assert abs(grads["document_tower.projection.2.weight"].norm().item() - 0.135800555) < 1e-7
    st = orc.AdamState(lr=1e-3)
    avg, _ = orc.train_epoch(table, [batch] * 3, params, st, margin)
    assert abs(avg - 0.537761847) < 1e-7
    assert abs(params["query_tower.projection.0.weight"][0, 0].item() - 0.253004819) < 1e-7


@pytest.mark.parametrize("name", ["kat0.npz", "step_mid.npz", "step_d384.npz"])
def test_full_step_matches_reference(name):
    g = load_golden(name)
    table = T(g["table"])
    batches = _batches(g)
    margin, lr = float(g["margin"]), float(g["lr"])
    params = _params(g, "init/")
    loss, grads, (q, p, n) = orc.loss_and_grads(table, batches[0], params, margin)
    for r, (ids, mask) in zip("qpn", batches[0]):
        np.testing.assert_allclose(orc.gather_pool_normalize(table, ids, mask).numpy(), g[f"fwd/pooled_{r}"], atol=TOL, rtol=0)
    np.testing.assert_allclose(q.numpy(), g["fwd/q"], atol=TOL, rtol=0)
    np.testing.assert_allclose(p.numpy(), g["fwd/p"], atol=TOL, rtol=0)
    np.testing.assert_allclose(n.numpy(), g["fwd/n"], atol=TOL, rtol=0)
    assert abs(loss.item() - float(g["fwd/loss"])) < TOL
    np.testing.assert_allclose(orc.cosine_distance(q, p).numpy(), g["fwd/pos_dist"], atol=TOL, rtol=0)
    for k in orc.PARAM_KEYS:
        np.testing.assert_allclose(grads[k].numpy(), g["grad/" + k], atol=TOL, rtol=0)
    # closed-form loss gradient == autograd of the reference
    _, _, _, dq, dp, dn = orc.triplet_loss_closed_form(q, p, n, margin)
    np.testing.assert_allclose(dq.numpy(), g["bwd/dq"], atol=TOL, rtol=0)
    np.testing.assert_allclose(dp.numpy(), g["bwd/dp"], atol=TOL, rtol=0)
    np.testing.assert_allclose(dn.numpy(), g["bwd/dn"], atol=TOL, rtol=0)
    # epoch: hand-written Adam == torch.optim.Adam driven by the reference train_epoch
    st = orc.AdamState(lr=lr)
    avg, trace = orc.train_epoch(table, batches, params, st, margin)
    np.testing.assert_allclose(trace, g["epoch/loss_trace"], atol=TOL, rtol=0)
    assert abs(avg - float(g["epoch/avg_loss"])) < TOL
    for k in orc.PARAM_KEYS:
        assert_params_close(params[k].numpy(), g["final/" + k], lr, len(batches))


def test_pool_edges():
    g = load_golden("pool_edges.npz")
    out = orc.gather_pool_normalize(T(g["table"]), T(g["ids"]), T(g["mask"]))
    np.testing.assert_allclose(out.numpy(), g["out"], atol=TOL, rtol=0)
    assert np.all(out.numpy()[2] == 0)  # all-padding row: clamp(1e-9) then 1e-12 normalise -> zeros
    np.testing.assert_array_equal(orc.embed_lookup(T(g["table"]), T(g["ids"])).numpy(), g["hidden"])  # bit-exact gather
    np.testing.assert_allclose(orc.pool_normalize(T(g["hidden"]), T(g["mask"])).numpy(), g["out"], atol=TOL, rtol=0)


def test_loss_edges():
    g = load_golden("loss_edges.npz")
    q, p, n = T(g["q"]), T(g["p"]), T(g["n"])
    m = float(g["margin"])
    assert abs(orc.triplet_loss(q, p, n, m).item() - float(g["loss"])) < TOL
    loss, pd, nd, dq, dp, dn = orc.triplet_loss_closed_form(q, p, n, m)
    assert abs(loss.item() - float(g["loss"])) < TOL
    np.testing.assert_allclose(pd.numpy(), g["pos_dist"], atol=TOL, rtol=0)
    np.testing.assert_allclose(nd.numpy(), g["neg_dist"], atol=TOL, rtol=0)
    for a, k in ((dq, "dq"), (dp, "dp"), (dn, "dn")):
        scale = max(1.0, np.abs(g[k]).max())
        np.testing.assert_allclose(a.numpy() / scale, g[k] / scale, atol=TOL, rtol=1e-5)


@pytest.mark.parametrize("name", ["bert_mode.npz", "bert_minilm.npz"])  # 2 layers / hidden 32; MiniLM-L6 shape (round 5)
def test_bert_mode_dense_pool(name):
    g = load_golden(name)
    pooled = orc.pool_normalize(T(g["hidden"]), T(g["mask"]))
    out = orc.projection(pooled, T(g["w1"]), T(g["b1"]), T(g["w2"]), T(g["b2"]))
    np.testing.assert_allclose(out.numpy(), g["out"], atol=TOL, rtol=0)


def test_tower_forward_through_an_encoder_module():
    """`bert` mode of the oracle (bench.py's cpu_baseline_bert): the frozen encoder module stands where the table
    does and its first output is pooled (backend/model.py:51-56) — pinned by the reference's own tower output."""
    g = load_golden("bert_mode.npz")

    class Encoder(torch.nn.Module):
        def forward(self, input_ids=None, attention_mask=None):
            assert torch.equal(input_ids, T(g["ids"])) and torch.equal(attention_mask, T(g["mask"]))
            return (T(g["hidden"]),)

    out = orc.tower_forward(Encoder(), T(g["ids"]), T(g["mask"]), T(g["w1"]), T(g["b1"]), T(g["w2"]), T(g["b2"]))
    np.testing.assert_allclose(out.numpy(), g["out"], atol=TOL, rtol=0)


def test_create_triplets_matches_reference():
    g = load_golden("triplets.npz")
    for case in range(6):  # cases 4-5 hold empty passages (data.py:126)
        qids = g[f"case{case}/query_ids"].tolist()
        rng = random.Random(int(g[f"case{case}/seed"]))
        assert orc.create_triplet_indices(qids, rng, g[f"case{case}/positive_nonempty"].tolist()) == g[f"case{case}/neg_idx"].tolist()


def test_adam_restatement_equals_torch_adam():
    torch.manual_seed(0)
    p0 = {k: torch.randn(5, 3) for k in ("a", "b")}
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in p0.items()}
    opt = torch.optim.Adam(ref.values(), lr=1e-2)
    mine = {k: v.clone() for k, v in p0.items()}
    st = orc.AdamState(lr=1e-2)
    for step in range(5):
        grads = {k: torch.randn(5, 3) for k in p0}
        for k in ref:
            ref[k].grad = grads[k].clone()
        opt.step()
        orc.adam_step(mine, grads, st)
    for k in p0:
        np.testing.assert_allclose(mine[k].numpy(), ref[k].detach().numpy(), atol=1e-7, rtol=0)


def test_accumulation_epochs_match_reference_train_epoch_optimized():
    """tests/golden/step_accum.npz: the reference's own `train_epoch_optimized` (backend/training.py:66-133) run in
    fp32 on the CPU over two epochs of five batches with accumulation_steps = 2 — the fifth batch's gradient is still
    in .grad when epoch 2 starts (training.py:99-102), so epoch 2's first optimizer step applies three of them."""
    g = load_golden("step_accum.npz")
    table, batches = T(g["table"]), _batches(g)
    k, n_epochs = int(g["accumulation_steps"]), int(g["n_epochs"])
    assert (len(batches), k, n_epochs) == (5, 2, 2)
    params = _params(g, "init/")
    st = orc.AdamState(lr=float(g["lr"]))
    carry, trace = None, []
    for e in range(n_epochs):
        avg, tr, carry = orc.train_epoch_accum(table, batches, params, st, float(g["margin"]), k, carry=carry)
        trace += tr
        assert abs(avg - float(g["epoch/avg_loss"][e])) < TOL
        for key in orc.PARAM_KEYS:
            assert_params_close(params[key].numpy(), g[f"epoch{e}/" + key], float(g["lr"]), st.step)
    assert st.step == int(g["optimizer_steps"]) == 4
    np.testing.assert_allclose(trace, g["epoch/loss_trace"], atol=TOL, rtol=0)
    for key in orc.PARAM_KEYS:  # what no optimizer step has consumed yet: the last batch's gradient / k
        np.testing.assert_allclose(carry[key].numpy(), g["carry/" + key], atol=TOL, rtol=0)
