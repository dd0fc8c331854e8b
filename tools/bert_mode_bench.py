#!/usr/bin/env python3
"""`bert` encoder mode (what backend/model.py:51-52 executes) with a MiniLM-L6-shaped frozen encoder (random weights:
no checkpoint offline): (a) the module path — three encoder passes per step, the reference's loop (training.py:36-53);
(b) the fused trainer — encoder once per text into the frozen-feature cache, then cache-fed epochs.
Prints triplets/s of each and the one-off cache-build time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformers
from two_towers_overlords_amd.data import TokenTripletLoader
from two_towers_overlords_amd.model import TripletLoss, TwoTowersModel
from two_towers_overlords_amd.training import FusedAdam, TripletTrainer, train_epoch

N, B, Sq, Sp, P = int(os.environ.get("N", 32768)), 2048, 32, 128, 384
torch.manual_seed(0)
cfg = transformers.BertConfig(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12,
                              intermediate_size=1536, max_position_embeddings=512)
bert = transformers.BertModel(cfg).eval()
gen = torch.Generator().manual_seed(1)
q = torch.randint(1000, 30000, (N, Sq), generator=gen).int()
p = torch.randint(1000, 30000, (N, Sp), generator=gen).int()
ql = torch.randint(4, Sq + 1, (N,), generator=gen).int()      # queries: 4..32 tokens
pl = torch.randint(32, Sp + 1, (N,), generator=gen).int()     # passages: 32..128 tokens
q = q * (torch.arange(Sq).unsqueeze(0) < ql.unsqueeze(1))
p = p * (torch.arange(Sp).unsqueeze(0) < pl.unsqueeze(1))
qids = torch.arange(N) // 8
model = TwoTowersModel(projection_dim=P, tokenizer=lambda *a, **k: None, pretrained_model=bert, encoder_mode="bert").cuda()

# (a) module path, a few steps
la = TokenTripletLoader.from_tensors(q, ql, p, pl, qids, batch_size=B, device="cuda", seed=3, sampler="device", emit="ids3", pad="batch")
def pairs(it, n):
    for i, (ids, lens) in enumerate(it):
        if i >= n:
            break
        S = ids.shape[2]
        m = (torch.arange(S, device="cuda").view(1, 1, S) < lens.unsqueeze(-1)).long()
        yield tuple((ids[r].long(), m[r]) for r in range(3))
opt, crit = FusedAdam(model.parameters(), lr=1e-4), TripletLoss(0.3)
train_epoch(model, pairs(la, 1), crit, opt, log_wandb=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
steps = 4
train_epoch(model, pairs(la, steps), crit, opt, log_wandb=False)
torch.cuda.synchronize(); ta = time.perf_counter() - t0
print(f"module path (3 encoder passes / step, B={B}): {steps * B / ta:,.0f} triplets/s ({ta / steps * 1e3:.1f} ms/step)")

# (b) fused: encoder once per text, then cache-fed epochs
lb = TokenTripletLoader.from_tensors(q, ql, p, pl, qids, batch_size=B, device="cuda", seed=3, sampler="device", emit="cache")
tr = TripletTrainer(model, max_batch=B, max_seq_len=Sp, margin=0.3, lr=1e-4)
ids_all, lens_all = lb.all_texts()
torch.cuda.synchronize(); t0 = time.perf_counter()
tr.build_feature_cache(ids_all, lens=lens_all, chunk=1024)
torch.cuda.synchronize(); tc = time.perf_counter() - t0
print(f"feature cache: {2 * N:,} texts through the frozen encoder once in {tc:.2f} s ({2 * N / tc:,.0f} texts/s)")
tr.calibrate(next(iter(lb)))
tr.train_epoch(lb)
torch.cuda.synchronize(); t0 = time.perf_counter()
E = 5
for _ in range(E):
    tr.train_epoch(lb)
torch.cuda.synchronize(); tb = time.perf_counter() - t0
print(f"cache-fed epochs: {E * N / tb:,.0f} triplets/s ({tb / E * 1e3:.1f} ms per epoch of {N:,} triplets)")
ep = 15
print(f"{ep} epochs of {N:,} triplets: module path {ep * N / (steps * B / ta):.1f} s; fused {tc + ep * tb / E:.2f} s "
      f"(= {ep * N / (tc + ep * tb / E):,.0f} triplets/s including the cache build)")
