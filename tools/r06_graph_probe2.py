#!/usr/bin/env python3
"""Round-6 diagnostic: which part of a text-store epoch, run before a step graph is captured, makes hipGraphLaunch crash?
   python tools/r06_graph_probe2.py MODE   (MODE: a '+'-joined subset of long,loader,modules,engine,gc,close)"""
import gc
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

mode = set(sys.argv[1].split("+")) if len(sys.argv) > 1 else set()
import test_gpu_model_api as t  # noqa: E402
import test_gpu_tokenizer as tk  # noqa: E402
import transformers  # noqa: E402

d = tempfile.mkdtemp()
vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"[unused{i}]" for i in range(20)] + list(tk.WORDS) + tk.PIECES
open(os.path.join(d, "vocab.txt"), "w").write("\n".join(vocab) + "\n")
tok = transformers.BertTokenizerFast.from_pretrained(d)

from two_towers_overlords_amd.data import TextColumn, TripletDataLoader  # noqa: E402
from two_towers_overlords_amd.model import TripletLoss  # noqa: E402
from two_towers_overlords_amd.training import train_epoch  # noqa: E402

rng = random.Random(11)
items = tk._marco_like(60, 300, rng)
if "long" in mode:
    items[17]["positive"] = " ".join(rng.choice(tk.WORDS) for _ in range(150))
ds = tk._DS(items)
crit = TripletLoss(margin=0.3)
for fused in ([False] if "modules" in mode else []) + ([None] if "engine" in mode else []):
    model, _ = tk._model(tok, 32, 5)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    random.seed(22)
    torch.manual_seed(22)
    if "loader" in mode:
        batches = TripletDataLoader(ds, batch_size=64, num_workers=0)
    else:
        batches = []
        BS = 32 if "b32" in mode else 64
        for s in range(0, len(items), BS):
            chunk = items[s: s + BS]
            neg = [(i * 7 + 3) % len(chunk) for i in range(len(chunk))]
            batches.append(([c["query"] for c in chunk], [c["positive"] for c in chunk], [chunk[j]["positive"] for j in neg]))
    if "pretok" in mode:  # the same texts as (input_ids, attention_mask) blocks: the role-packing path, no text store
        pre = []
        for q, p_, n_ in batches:
            enc = [tok(list(x), padding=True, truncation=True, return_tensors="pt", max_length=512) for x in (q, p_, n_)]
            pre.append(tuple((e["input_ids"].cuda(), e["attention_mask"].cuda()) for e in enc))
        batches = pre
    if "nopin" in mode:
        torch.Tensor.pin_memory = lambda self, *a, **k: self
    for _ in range(int(os.environ.get("PROBE_EPOCHS", "3"))):
        train_epoch(model, batches, crit, opt, log_wandb=False, fused=fused, text_cache=(False if "inline" in mode else None))
    if "close" in mode and getattr(opt, "_tt_route", None) is not None:
        opt._tt_route.trainer.engine.close()
    del model, opt
if "gc" in mode:
    gc.collect()
torch.cuda.synchronize()
from two_towers_overlords_amd import engine as _eng, trainer_graphs as _tg  # noqa: E402
import traceback  # noqa: E402

_close = _eng.HotPathEngine.close


def traced_close(self):
    if getattr(self, "_h", None):
        print("CLOSE engine", hex(id(self)), "capturing=", torch.cuda.is_current_stream_capturing(), "from:",
              " <- ".join(f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-6:-1]), flush=True)
    return _close(self)


_eng.HotPathEngine.close = traced_close
_g = _tg.GraphReplayMixin._graph


def traced_graph(self, a, b):
    new = (a, b) not in self._graphs
    if new:
        print("CAPTURE begin", a, b, flush=True)
    out = _g(self, a, b)
    if new:
        print("CAPTURE end", a, b, flush=True)
    return out


_tg.GraphReplayMixin._graph = traced_graph
from two_towers_overlords_amd import trainer as _tr  # noqa: E402
_tr.TripletTrainer._graph = traced_graph
print("before graph test", flush=True)
t.test_graph_replay_equals_eager_and_oracle()
print("ok", sorted(mode), flush=True)
