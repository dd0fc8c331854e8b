#!/usr/bin/env python3
"""bench.py — triplets/sec of the full training step (3 encodes + triplet loss +
backward + Adam [+ gradient all-reduce]) of the two-tower hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], the shape the metric is quoted on): synthetic
(q, pos, neg) token-id triplets, seq_len 128 full length, MiniLM-shaped table
V=30522 x D=384 (N(0, 0.02^2), seed 1), proj_dim 384, 2048 triplets per GPU per
step (weak scaling), `embed` encoder mode, fp32.  Inputs (ids) are resident in
HBM before the timed region.  One JSON line on rank 0.

What one timed step is: the trainer is a two-stage pipeline (the weight-independent
gather+pool of batch i+2 runs on a side stream beside the projection / loss /
backward / Adam of batch i).  The pipeline is filled during warm-up and drained
after the timed region, so the K timed steps are K steady-state steps: exactly K
gather+pool launches and K fwd+bwd+Adam chains are enqueued and completed between
the two barrier + synchronize brackets (the gathers consumed by the first steps
were produced by warm-up; the last steps' gathers are consumed by the drain).
`--e2e` times a whole epoch instead, batches produced by TokenTripletLoader, fill
and drain included.

Extra objects: `roofline` (K1 gather+pool kernel, HBM bound, live HIP-event
timing), `cpu_baseline` (the CPU oracle — a torch-CPU restatement of the
reference loop — on a bounded sample of the same workload, `embed` mode),
`cpu_baseline_bert` (the same loop with the frozen MiniLM-L6-shaped encoder the
reference actually executes, backend/model.py:51-52) and `secondary` (the same
measurement at BASELINE.json configs[3], the one configuration whose table
cannot be cache-resident) — the last three on rank 0 at N=1 only.

`--gpus N` with N > 1 from a plain shell starts itself under
`python -m torch.distributed.run` (one rank per GPU) as a child process, before
anything in this process has touched the GPU, and exits with the child's code.
"""
import argparse
import copy
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: V, D, P, B per GPU, S
    "c2": dict(V=30522, D=384, P=384, B=2048, S=128, label="configs[1]: MiniLM-L6 table V=30522 d=384, proj 384, batch 2048/GPU, seq 128"),
    "c3": dict(V=30522, D=384, P=512, B=1024, S=128, label="configs[2]: one GPU's share of batch 8192 over 8 GPUs (1024/GPU), proj 512, seq 128"),
    "c4": dict(V=250000, D=768, P=768, B=16384, S=256, label="configs[3]: synthetic V=250k d=768 proj 768 batch 16384/GPU seq 256 (HBM stress)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
# MI355X_MICROARCH.md "Indexed rows: gather into LDS": rows shared by every workgroup and served by the XCD's L2 are
# gathered at 66-73 GB/s per CU = 16.8-18.8 TB/s chip-wide — the ceiling that binds a gather whose table is cache-resident
L2_SHARED_ROWS_GBS = (16800.0, 18800.0)
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md "L2 (per XCD)": 4 MiB per XCD, ~34.5 TB/s aggregate
CONFIG0 = dict(V=30522, D=384, P=64, B=256, S=128)  # BASELINE.json configs[0]: --batch-size 256 --projection-dim 64 (reference README.md:64-75)


def k1_algorithmic_bytes(n_texts, S, D):
    """per text n_valid*D*4 (table rows) + S*4 (ids as int32) + D*4 (output)."""
    return n_texts * (S * D * 4 + S * 4 + D * 4)


def k2_mfma_util(config):
    """MFMA utilisation of the projection GEMMs (K2) at this configuration from the newest committed PMC summary
    (profiles/rNN_k2_mfma_util.txt: SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE per kernel, one-stream runs — counters are
    not collected in a timed run).  The second bound the step lives under: exact-fp32 MFMA, 157.3 TFLOP/s."""
    import glob
    import re

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_k2_mfma_util.txt")))
    if not paths:
        return None
    want = {"c2": "configs[1] B=2048", "c3": "configs[2] one GPU", "c4": "configs[3]"}.get(config, config)
    kernels, on = {}, False
    for line in open(paths[-1]):
        if line.startswith("=="):
            on = want is not None and want in line and "fork" not in line
            continue
        m = re.match(r"\s*(gemm\S+(?:, [^>]*)*>?)\s+mfma_util\s+([0-9.]+)", line)
        if on and m:
            kernels[m.group(1).strip()] = float(m.group(2))
    if not kernels:
        return None
    return {"bound": "mfma", "peak": 157.3, "unit": "TFLOP/s", "dtype": "f32 (v_mfma_f32_32x32x2_f32, exact)",
            "mfma_util": kernels, "mfma_util_min": min(kernels.values()), "mfma_util_max": max(kernels.values()),
            "source": os.path.relpath(paths[-1], ROOT) + ": rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, one stream"}


def make_ids(V, B, S, n_batches, seed, device, zipf=False):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_batches):
        if zipf:
            u = torch.rand(3, B, S, generator=g, dtype=torch.float64)
            ids = (torch.exp(u * torch.log(torch.tensor(float(V)))) - 1).long().clamp_(0, V - 1)
        else:
            ids = torch.randint(0, V, (3, B, S), generator=g, dtype=torch.int64)
        out.append(ids.to(torch.int32).to(device))
    return out


def cpu_baseline(cfg, budget_s=20.0, max_batches=64):
    """Oracle train_epoch (torch-CPU restatement of backend/training.py:26-63) on all host cores."""
    from oracle import two_towers_oracle as orc

    # threads actually used: the cores this process may run on, capped at the GPU box's
    # per-GPU CPU share (16) — oversubscribing a 256-thread host makes torch-CPU far slower
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    V, D, P, S = cfg["V"], cfg["D"], cfg["P"], cfg["S"]
    B = min(cfg["B"], 2048)
    table = orc.synthetic_table(V, D, seed=1)
    doc_table = table.clone()  # the reference holds one frozen model per tower (backend/model.py:84-85)
    params = orc.init_params(D, P, seed=2)
    st = orc.AdamState(lr=1e-3)
    batch = orc.synthetic_batch(B, S, V, seed=1234)
    orc.train_epoch(table, [batch], params, st, 0.3, doc_table=doc_table)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        orc.train_epoch(table, [batch], params, st, 0.3, doc_table=doc_table)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= max_batches:
            break
    return {
        "value": n * B / el, "unit": "triplets/s", "cores": cores, "kind": "port",
        "sample": f"{n} batches of {B} triplets (S={S}, D={D}, P={P}, V={V}, embed mode, fp32) in {el:.1f} s, "
                  f"oracle/two_towers_oracle.train_epoch, torch {torch.__version__} CPU, {cores} threads",
    }


def cpu_baseline_bert(cfg, budget_s=10.0):
    """The reference's EXECUTED CPU path (backend/model.py:51-52: the frozen 6-layer MiniLM between gather and pool,
    SURVEY.md F2/F3) on the same host threads: oracle train_epoch with a random-weight MiniLM-L6-shaped
    `transformers.BertModel` per tower (weights are not in the image; the arithmetic and its cost do not depend on
    their values), eval mode like the reference's CPU loop.  Bounded to about `budget_s` seconds."""
    from oracle import two_towers_oracle as orc

    try:
        from transformers import BertConfig, BertModel
    except Exception as e:  # transformers missing on the box: say so instead of inventing a number
        return {"value": None, "unit": "triplets/s", "kind": "port", "sample": f"not measured: {type(e).__name__}: {e}"}
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    V, D, P, S = cfg["V"], cfg["D"], cfg["P"], cfg["S"]
    B = 64
    bc = BertConfig(vocab_size=V, hidden_size=D, num_hidden_layers=6, num_attention_heads=12, intermediate_size=4 * D,
                    max_position_embeddings=512)
    torch.manual_seed(3)
    enc_q, enc_d = BertModel(bc).eval(), BertModel(bc).eval()  # one frozen model PER tower (backend/model.py:84-85)
    params = orc.init_params(D, P, seed=2)
    st = orc.AdamState(lr=1e-3)
    batch = orc.synthetic_batch(B, S, V, seed=1234)
    n, t0 = 0, time.perf_counter()
    while True:
        orc.train_epoch(enc_q, [batch], params, st, 0.3, doc_table=enc_d)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 16:
            break
    return {
        "value": n * B / el, "unit": "triplets/s", "cores": cores, "kind": "port",
        "sample": f"{n} batches of {B} triplets (S={S}, D={D}, P={P}, V={V}, bert mode: random-weight 6-layer/12-head/"
                  f"{4 * D}-ffn BertModel per tower under no_grad, eval mode, fp32) in {el:.1f} s, first batch included, "
                  f"oracle/two_towers_oracle.train_epoch, torch {torch.__version__} CPU, {cores} threads",
    }


def run_e2e(args, trainer, cfg, device, rank, world, barrier):
    """One epoch of `--steps` batches through TokenTripletLoader (sampler="device") and
    TripletTrainer.train_epoch: per batch the loader shuffles, draws the in-batch negatives and gathers
    the [3,B,S] token ids on the device; the trainer's pipeline fill and drain and the epoch-end loss
    read-back are inside the timed region.  Warm-up: `--warmup` batches of a first, untimed pass."""
    from two_towers_overlords_amd.data import TokenTripletLoader

    V, B, S = cfg["V"], cfg["B"], cfg["S"]
    n_items = args.steps * B
    g = torch.Generator().manual_seed(4242 + rank)
    mk = lambda: torch.randint(0, V, (n_items, S), generator=g, dtype=torch.int32).to(device)
    lens = torch.full((n_items,), S, dtype=torch.int32, device=device)
    qids = torch.arange(n_items) // 4  # four passages per query
    loader = TokenTripletLoader.from_tensors(mk(), lens, mk(), lens.clone(), qids, batch_size=B, device=device,
                                             seed=11 + rank, sampler="device", emit=("rows3" if args.inputs == "ids3" else "rows2"))
    trainer.set_token_store(*loader.all_texts())  # [2N, S] token rows, resident; batches name their texts by row
    as_b = lambda it: it

    def first(n):
        for i, b in enumerate(loader):
            if i >= n:
                break
            yield b

    if args.warmup > 0:
        trainer.train_epoch(as_b(first(args.warmup)), global_batch_sizes=None)
    barrier()
    k1_ev = []
    trainer.k1_events = k1_ev
    t0 = time.perf_counter()
    loss = trainer.train_epoch(as_b(iter(loader)))  # ends with the epoch's one host<-device read (the loss)
    barrier()
    elapsed = time.perf_counter() - t0
    trainer.k1_events = None
    return elapsed, {"items": n_items, "batches": len(loader), "loss": loss, "k1_events": k1_ev}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="triplets per GPU per step (default: the config's)")
    ap.add_argument("--vocab", type=int, default=None,
                    help="diagnostic: rows of the embedding table (default: the config's) — how the step depends on whether a "
                         "column slice of the table fits an XCD's L2 beside the projection chain's operands")
    ap.add_argument("--zipf", action="store_true", help="Zipf(1) token ids instead of uniform")
    ap.add_argument("--ragged", action="store_true",
                    help="ragged prefix lengths (queries U{8..S}, documents U{32..S}; SURVEY.md §8d) instead of full length")
    ap.add_argument("--k1-only", action="store_true", help="time only the gather+pool kernel (diagnostic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the gather of batch i+1 after (not under) the projection kernels of batch i")
    ap.add_argument("--k1-blocks-per-cu", type=int, default=None)
    ap.add_argument("--graphs", action="store_true",
                    help="replay each step as one hipGraph (opt-in: slower on ROCm 7.2, see DESIGN.md)")
    ap.add_argument("--inputs", default="ids3", choices=["ids3", "index", "cache"],
                    help="ids3 (headline): three token-id gathers per triplet batch, the reference's three encodes. "
                         "index: negatives as in-batch indices of the positives (2 gathers). cache: rows of a "
                         "frozen-feature cache + negative indices (epochs >= 2 of real training).  The last two are "
                         "reported with their own metric name; they are not the BASELINE metric.")
    ap.add_argument("--loss", default="triplet", choices=["triplet", "inbatch"],
                    help="inbatch: all in-batch negatives (BASELINE.json configs[4]; B x B cosine matrix on MFMA), "
                         "2 gathers per batch; reported under its own metric name")
    ap.add_argument("--e2e", action="store_true",
                    help="time one whole epoch of --steps batches produced by TokenTripletLoader (device sampler: shuffle, "
                         "in-batch negative draws and the [3,B,S] id gathers happen per batch, on the device) instead of "
                         "cycling pre-staged batches; reported under its own metric name")
    ap.add_argument("--h2d", action="store_true",
                    help="PCIe-inclusive rate: every step's ids [3,B,S] int32 come from pinned HOST memory by a "
                         "non_blocking copy on a copy stream (what the reference's `.to(device)` of the tokenizer "
                         "output is, backend/model.py:48); reported under its own metric name")
    ap.add_argument("--simulate-allreduce-us", type=float, default=None,
                    help="one GPU only, diagnostic: run the data-parallel step (bucket reduce, exchange, Adam from the "
                         "bucket) with the all-reduce replaced by a kernel of this duration on the caller's stream (where a "
                         "synchronous-op RCCL collective runs) — how much of an N-GPU all-reduce's latency the step would "
                         "expose (RCCL's own CU use is not modelled)")
    ap.add_argument("--calibrate-k1", type=int, default=None, choices=[0, 1],
                    help="whether calibrate() also times the gather's form on this run's ids where the shape rule picks the "
                         "binned form (TripletTrainer.calibrate(k1_forms=...)); default: the trainer's rule — on for a table "
                         "beyond the Infinity Cache; the pick is in calibration.k1_form")
    ap.add_argument("--ids-form", default="packed", choices=["packed", "lens", "roles", "padids"],
                    help="diagnostic: how the engine path's batches carry their ids — packed int32 [3,B,S] (headline), the same "
                         "with full prefix lengths, or the tokenizer's form: three (int64 input_ids, int64 attention_mask) "
                         "blocks packed by the engine (what the drop-in route hands over)")
    ap.add_argument("--no-calibrate", action="store_true",
                    help="skip TripletTrainer.calibrate() in set-up (it times the two backward schedules on this device "
                         "and keeps the faster; its ~190 dry steps also bring the GPU out of its idle power state)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse the multi-process "
                         "path with several ranks sharing one GPU)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` object (the same measurement at configs[3], V=250k d=768 S=256 B=16384: "
                         "the one configuration whose table cannot be cache-resident; ~10 s extra on the default run)")
    ap.add_argument("--path", default="engine", choices=["engine", "module", "module-unfused"],
                    help="engine (headline): TripletTrainer on the fused engine.  module: the drop-in surface a user of the "
                         "reference gets by swapping the imports (INTEGRATION.md section 1) — training.train_epoch(model, "
                         "dataloader, TripletLoss, FusedAdam) on pre-tokenised int64 (input_ids, attention_mask) batches; it "
                         "routes its batch body to the fused engine.  module-unfused: the same call with fused=False — "
                         "TwoTowersModel.encode_* as autograd Functions over the same kernels, one multi-tensor Adam launch.  "
                         "Both reported under their own metric names")
    ap.add_argument("--optimizer", default="torch", choices=["torch", "fused"],
                    help="--path module*: torch = the reference's own `torch.optim.Adam(model.parameters(), lr)` "
                         "(backend/training.py:436) [default]; fused = this package's FusedAdam")
    ap.add_argument("--strings", default=None, choices=["lists", "columns", "loader"],
                    help="--path module: batches of STRINGS through a real BertTokenizerFast (the reference's batch form, "
                         "backend/training.py:36-42).  lists = plain list[str] batches; columns = the row-tagged lists this "
                         "package's TripletDataLoader yields, pre-built; loader = the TripletDataLoader itself "
                         "(shuffle + the reference's in-batch negative draws per batch).  Reports epoch 1 (tokenises) and "
                         "epochs >= 2 (the text store) separately")
    ap.add_argument("--epochs", type=int, default=None, help="--strings: epochs over the same --steps batches (1 cold + the rest warm; "
                    "default 3).  --run-training: num_epochs (default 15)")
    ap.add_argument("--run-training", action="store_true",
                    help="time a WHOLE run_training(...) call (backend/training.py:383-529) end to end: tokenizer, frozen-feature cache, "
                         "epochs, per-epoch validation, final test (bench_run_training.py); its own metric name")
    ap.add_argument("--pairs", type=int, default=100000, help="--run-training: (query, positive) pairs = max_samples")
    ap.add_argument("--proj", type=int, default=512, help="--run-training: projection_dim")
    ap.add_argument("--encoders", default="bert,embed", help="--run-training: encoder modes to run, comma-separated (bert = the reference's "
                    "executed forward with a MiniLM-L6-shaped frozen encoder; bert:fp16 / bert:bf16 = the same with the frozen encoder under "
                    "torch.autocast like the reference's GPU loop (NOT the parity path); embed = the word-embedding gather)")
    ap.add_argument("--text-cache", type=int, default=1, choices=[0, 1],
                    help="--strings: 0 = train_epoch(text_cache=False): the three tokenizer calls on every batch, as the reference does")
    ap.add_argument("--eval", action="store_true",
                    help="SURVEY §8 f2: time evaluate_model (backend/training.py:136-380) at the reference's two shapes — queries/s "
                         "end to end, the split, each device op alone with its roofline, the reference's per-query loop on the CPU "
                         "oracle beside it (bench_eval.py); its own metric name")
    ap.add_argument("--eval-repeats", type=int, default=3, help="--eval: timed calls per shape after one warm-up call (best is reported)")
    ap.add_argument("--debug-trace", default=None, choices=["10", "step"],
                    help="diagnostic: timing events inside the timed loop (every 10th step, or every step) printed to stderr")
    ap.add_argument("--debug-prewarm-ms", type=float, default=0.0, help="diagnostic: keep the GPU busy this long before warm-up")
    ap.add_argument("--debug-idle-ms", type=float, default=0.0, help="diagnostic: idle gap between warm-up and the timed region")
    ap.add_argument("--debug-no-gc-freeze", action="store_true", help="diagnostic: skip gc.freeze() before the timed loop")
    ap.add_argument("--debug-hostprof", action="store_true",
                    help="--strings: cProfile the last epoch's host side to stderr (the epoch is slower for it)")
    ap.add_argument("--tune", action="append", default=[], metavar="FIELD=VALUE",
                    help="a tt_tuning field for this run's engine (include/tt_hotpath.h), e.g. --tune k1_form=stream "
                         "--tune proj_tile=22; repeatable.  Speed only; recorded in config.tuning")
    ap.add_argument("--trainer", action="append", default=[], metavar="FIELD=VALUE",
                    help="a TrainerConfig field for this run's TripletTrainer (two-towers-overlords_amd/training.py), e.g. "
                         "--trainer join_carries_gather=false --trainer accumulation_steps=2; repeatable; recorded in "
                         "config.trainer.  (the older single-purpose flags below set the same fields)")
    ap.add_argument("--event-scope", type=int, default=None, choices=[1, 2, 3],
                    help="release scope of the events that order the gather stream and the caller's stream, and the engine's "
                         "two streams (1 no system-scope fence [default], 2 device-scope release, 3 HIP's default events)")
    ap.add_argument("--fence-at-loss", type=int, default=None, choices=[0, 1],
                    help="A/B: 1 = the gather stream's slot fence is the engine's loss / fork event, 0 = an event of its own at "
                         "the start of the step (TripletTrainer(fence_at_loss=...)); default: calibrate() decides")
    ap.add_argument("--lib", default=None,
                    help="diagnostic: load this build of libtt_hotpath.so instead of the in-tree one (same-box A/B of two builds "
                         "through tools/ab_tune.py)")
    ap.add_argument("--join-carries-gather", type=int, default=1, choices=[0, 1],
                    help="A/B: 1 [default] = in the fork backward the engine's join also carries the wait for the next batch's "
                         "gather (one wait packet per step on the caller's stream), 0 = a wait of its own ahead of Adam")
    ap.add_argument("--grad-slabs", type=int, default=0,
                    help="split-K slabs of the weight-gradient GEMMs (tt_engine_config.grad_slabs; 0 = the engine's choice)")
    ap.add_argument("--dp-split", default=None, choices=["0", "1", "auto"],
                    help="gradient exchange in one bucket (0), two parts (1: TripletTrainer(split_exchange=True)), or auto: "
                         "calibrate() times the data-parallel step both ways on the fabric at hand and keeps the faster "
                         "(default: auto with N > 1 ranks, 0 for the one-GPU stand-in)")
    ap.add_argument("--rendezvous-timeout", type=float, default=180.0,
                    help="N > 1: seconds allowed for init_process_group and for the first collective before the rank prints its "
                         "state and exits non-zero")
    ap.add_argument("--trace-steps", type=int, default=24,
                    help="steps of the separate, event-traced pass behind the timed region that `dp_breakdown` is computed from")
    ap.add_argument("--no-cpu-baseline-bert", action="store_true",
                    help="skip `cpu_baseline_bert` (the reference's executed CPU path: frozen MiniLM-shaped encoder, ~10 s)")
    args = ap.parse_args()
    args.tuning = {}
    if args.event_scope:
        args.tuning["event_scope"] = args.event_scope
    for kv in args.tune:
        k, _, v = kv.partition("=")
        args.tuning[k] = v if not v.lstrip("-").isdigit() else int(v)

    def _val(v):
        lv = v.lower()
        if lv in ("none", "null"):
            return None
        if lv in ("true", "false"):
            return lv == "true"
        for cast in (int, float):
            try:
                return cast(v)
            except ValueError:
                pass
        return v

    args.trainer_cfg = {kv.partition("=")[0]: _val(kv.partition("=")[2]) for kv in args.trainer}
    if args.lib:
        from two_towers_overlords_amd import _lib as _ttl
        _ttl.LIB_PATH = os.path.abspath(args.lib)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))  # nothing above has touched the GPU (device_count() does not initialise it)
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP library is the only implementation of the hot path")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev  # gloo rehearsal: several ranks may share one GPU
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        import datetime

        to = datetime.timedelta(seconds=args.rendezvous_timeout)
        with watchdog(args.rendezvous_timeout + 30, "init_process_group", rank):
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=to)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=to)
        with watchdog(args.rendezvous_timeout + 30, "first collective (all_reduce of one float)", rank):
            probe = torch.ones(1, device=device)
            dist.all_reduce(probe)
            torch.cuda.synchronize(device)
            assert int(probe.item()) == world

    if args.run_training:
        from bench_run_training import run_training_bench

        if world > 1:
            sys.exit("--run-training is a single-process measurement")
        print(json.dumps(run_training_bench(args, device)))
        return
    if args.eval:
        from bench_eval import run_eval

        if world > 1:
            sys.exit("--eval is a single-process measurement")
        print(json.dumps(run_eval(args, device, CONFIGS)))
        return
    if args.path != "engine":
        if world > 1:
            sys.exit("--path module is a single-process measurement")
        from bench_module import run_module_path  # the drop-in surface's own measurements (bench_module.py)

        print(json.dumps(run_module_path(args, device, CONFIGS, make_ids)))
        return
    default_line = (args.config == "c2" and args.inputs == "ids3" and args.loss == "triplet" and not args.k1_only
                    and not args.e2e and not args.h2d and args.simulate_allreduce_us is None and not args.batch
                    and not args.vocab and not args.lib and not args.zipf and not args.ragged and not args.tuning and args.dp_split is None and args.ids_form == "packed" and not args.trainer
                    and args.fence_at_loss is None and not args.grad_slabs)
    # on the default line the GPU legs run back to back (headline, then `secondary`) and the ~30 s of CPU baselines come
    # last: a GPU that has idled through them starts the next GPU leg in its power ramp (docs/LOG_r04.md §5 "Cold start")
    args.cpu_baseline_later = bool(rank == 0 and world == 1 and default_line and not args.no_secondary)
    result = run_config(args, world, rank, device)
    if rank == 0 and world == 1 and default_line:
        # the headline line must come out whatever happens to the extras: a failure in one of them is reported in its place
        if not args.no_secondary:
            try:
                result["secondary"] = secondary_c4(args, world, rank, device)
            except Exception as e:  # e.g. not enough free HBM for the 768 MB table + workspace on a shared device
                result["secondary"] = {"error": f"{type(e).__name__}: {e}"}
        if args.cpu_baseline_later and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(CONFIGS[args.config])
        if not args.no_cpu_baseline:  # the reference's own CPU-runnable case (BASELINE.json configs[0]: B = 256, P = 64)
            try:
                result["cpu_baseline_configs0"] = cpu_baseline(CONFIG0, budget_s=6.0, max_batches=400)
            except Exception as e:
                result["cpu_baseline_configs0"] = {"value": None, "unit": "triplets/s", "kind": "port",
                                                   "sample": f"not measured: {type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and not args.no_cpu_baseline_bert:
            try:
                result["cpu_baseline_bert"] = cpu_baseline_bert(CONFIGS[args.config])
            except Exception as e:
                result["cpu_baseline_bert"] = {"value": None, "unit": "triplets/s", "kind": "port",
                                               "sample": f"not measured: {type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


class watchdog:
    """A bounded wait for a step that can hang when a peer is missing (rendezvous, the first collective): if the body has
    not finished after `seconds`, print this rank's state and leave with a non-zero code — from a timer thread, by
    os._exit: nothing is exec'ed and no process is replaced, the launcher sees a failed child and tears the others down."""

    def __init__(self, seconds: float, what: str, rank: int):
        import threading

        def fire():
            print(f"bench.py rank {rank}: '{what}' did not finish within {seconds:.0f} s — MASTER_ADDR="
                  f"{os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')} WORLD_SIZE="
                  f"{os.environ.get('WORLD_SIZE')} LOCAL_RANK={os.environ.get('LOCAL_RANK')} "
                  f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}; giving up", file=sys.stderr, flush=True)
            os._exit(3)

        self.t = threading.Timer(seconds, fire)
        self.t.daemon = True

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def self_launch(args) -> int:
    """`bench.py --gpus N` (N > 1) from a plain shell: one rank per GPU under torch.distributed.run, started as a
    CHILD process (a process that has initialised the GPU must never exec another program on this pool); the rank-0
    JSON line arrives on the inherited stdout.  Returns the child's exit code."""
    ndev = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if args.backend == "nccl" and args.gpus > ndev:
        print(f"bench.py --gpus {args.gpus}: this node shows {ndev} GPU(s); RCCL needs one GPU per rank "
              "(--backend gloo rehearses the multi-process flow with ranks sharing a GPU)", file=sys.stderr)
        return 2
    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool's host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def secondary_c4(args, world, rank, device) -> dict:
    """BASELINE.json configs[3] (V=250k, d=768, proj 768, S=256, B=16384: a 768 MB table, three times the Infinity
    Cache — the configuration whose table cannot be cache-resident), measured in this same run with the same code path:
    step rate, K1 in the loop and K1 alone.  `traffic` comes from the newest committed PMC passes of this kernel
    (profiles/rNN_k1_traffic_c4.json; `traffic_source` says so): PMC counters are not collected in a timed run."""
    a = copy.copy(args)
    a.config, a.steps, a.warmup, a.no_cpu_baseline = "c4", 20, 6, True
    a.tuning = {}
    # ~130 steps of set-up (0.8 s) before the 20 timed ones: the first ~70 steps after an idle gap run 10-20 % slower
    a.n_distinct, a.calibrate_steps, a.calibrate_rounds, a.iso_launches = 2, 24, 2, 8
    torch.cuda.empty_cache()
    r = run_config(a, world, rank, device)
    torch.cuda.empty_cache()
    keep = ("metric", "value", "unit", "steps", "warmup", "effective_warmup_steps", "ms_per_step", "dtype", "config",
            "roofline", "calibration")
    out = {k: r[k] for k in keep}
    out["roofline_hbm_triplets_per_s"] = HBM_PEAK_GBS * 1e9 / (3 * (CONFIGS["c4"]["S"] * CONFIGS["c4"]["D"] * 4
                                                                   + CONFIGS["c4"]["S"] * 4 + CONFIGS["c4"]["D"] * 4))
    return out


def breakdown(trace):
    """Per-step averages (ms) from TripletTrainer.step_trace: chain = forward + loss + backward (+ slab reduce), exchange,
    adam, and `other` = what is left of the step (stream idle between Adam and the next step's first kernel: event
    waits, host gaps) — so the four sum to `traced_ms_per_step` by construction."""
    rows = [t for t in trace if all(e is not None for e in t)]
    if len(rows) < 3:
        return None
    rows = rows[1:]  # the first traced step still runs behind untraced ones
    n = len(rows) - 1
    chain = sum(rows[i][0].elapsed_time(rows[i][1]) for i in range(n)) / n
    exch = sum(rows[i][1].elapsed_time(rows[i][2]) for i in range(n)) / n
    adam = sum(rows[i][2].elapsed_time(rows[i][3]) for i in range(n)) / n
    step = sum(rows[i][0].elapsed_time(rows[i + 1][0]) for i in range(n)) / n
    return {"chain_ms": chain, "exchange_ms": exch, "adam_ms": adam, "other_ms": step - chain - exch - adam,
            "traced_ms_per_step": step, "steps": n}


def run_config(args, world, rank, device) -> dict:
    """Set-up, warm-up, K timed steady-state steps and the result object for one configuration."""
    n_distinct = getattr(args, "n_distinct", 8)
    import two_towers_overlords_amd as pkg
    from two_towers_overlords_amd.training import TripletTrainer

    pkg.load()
    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    if args.vocab:  # diagnostic: another table height (config.vocab records it; not a BASELINE workload)
        cfg["V"] = args.vocab
        cfg["label"] += f" [diagnostic: vocab {args.vocab}]"
    V, D, P, B, S = cfg["V"], cfg["D"], cfg["P"], cfg["B"], cfg["S"]

    gt = torch.Generator().manual_seed(1)
    table = (torch.randn(V, D, generator=gt) * 0.02).to(device)
    from two_towers_overlords_amd.engine import flat_param_slices
    gp = torch.Generator().manual_seed(2)
    params = {}
    for name, _, shape in flat_param_slices(D, P):
        bound = 1.0 / (shape[-1] if len(shape) == 2 else (D if ".0." in name else P)) ** 0.5
        params[name] = ((torch.rand(*shape, generator=gp) * 2 - 1) * bound).to(device)
    sim = args.simulate_allreduce_us is not None and world == 1
    if sim:
        class TripletTrainer(TripletTrainer):  # noqa: F811  (diagnostic subclass: the exchange is a timed stand-in)
            def __init__(self, *a, **kw):
                super().__init__(*a, exchange=True, **kw)  # the data-parallel step's code path on one rank
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(1000); torch.cuda.synchronize()
                e0.record(); torch.cuda._sleep(10_000_000); e1.record(); e1.synchronize()
                self._cycles = int(args.simulate_allreduce_us * 10_000_000 / (e0.elapsed_time(e1) * 1e3))

            def _all_reduce(self, bucket):  # on the caller's stream, like a synchronous-op collective
                if self._cycles > 0:
                    torch.cuda._sleep(self._cycles)

            def _all_reduce_parts(self, parts, bucket=None):  # --dp-split: each part costs a whole exchange (latency-bound sizes)
                if self._cycles > 0:
                    torch.cuda._sleep(self._cycles)

    tkw = dict(margin=0.3, lr=1e-3,
                             overlap_streams=(False if (args.no_overlap or args.k1_only) else None),
                             k1_blocks_per_cu=args.k1_blocks_per_cu, use_graphs=bool(args.graphs), loss=args.loss,
                             tuning=(args.tuning or None),
                             split_exchange=({"0": False, "1": True, "auto": None}[args.dp_split] if args.dp_split is not None
                                             else (None if world > 1 else False)),
                             fence_at_loss=(None if args.fence_at_loss is None else bool(args.fence_at_loss)),
                             grad_slabs=args.grad_slabs, join_carries_gather=bool(args.join_carries_gather),
                             pack_roles=(args.ids_form == "roles"))
    tkw.update(getattr(args, "trainer_cfg", {}))  # --trainer FIELD=VALUE wins over the single-purpose flags
    trainer = TripletTrainer(table=table, params=params, max_batch=B, max_seq_len=S, **tkw)
    if trainer.use_graphs:
        trainer.eager_every = 8   # every 8th step is launched kernel by kernel, with HIP events around its K1
        trainer.k1_event_every = 1
    id_batches = make_ids(V, B, S, n_distinct, seed=1234 + rank, device=device, zipf=args.zipf)
    batches = [(ids, None, None) for ids in id_batches]  # full-length: no mask, no lens
    valid_tokens = 3 * B * S
    if args.ragged:
        gl = torch.Generator().manual_seed(4321 + rank)
        lens_list = [torch.stack([torch.randint(8, S + 1, (B,), generator=gl), torch.randint(min(32, S), S + 1, (B,), generator=gl),
                                  torch.randint(min(32, S), S + 1, (B,), generator=gl)]).int() for _ in range(n_distinct)]
        valid_tokens = sum(int(l.sum()) for l in lens_list) / n_distinct
        batches = [(ids, None, l.to(device)) for ids, l in zip(id_batches, lens_list)]
    if args.ids_form == "padids" and args.ragged:  # the ragged lengths marked inside the ids (TT_PAD_ID) instead of a lens tensor
        pos = torch.arange(S, device=device).view(1, 1, S)
        batches = [(torch.where(pos < l.unsqueeze(2), ids, torch.full((), -(2 ** 31), dtype=ids.dtype, device=device)), None, None)
                   for ids, _, l in batches]
    if args.ids_form == "lens" and not args.ragged:
        full = torch.full((3, B), S, dtype=torch.int32, device=device)
        batches = [(ids, None, full) for ids in id_batches]
    if args.ids_form == "roles":
        from two_towers_overlords_amd.training import TripletBatch as _TBr

        ones = torch.ones(B, S, dtype=torch.int64, device=device)
        batches = [_TBr(role_blocks=[(ids[r].long(), ones) for r in range(3)]) for ids in id_batches]
    if args.loss == "inbatch":
        from two_towers_overlords_amd.training import TripletBatch

        args.inputs = "ids2"
        batches = [TripletBatch(ids=ids[:2].contiguous()) for ids in id_batches]
    elif args.inputs != "ids3":
        from two_towers_overlords_amd.training import TripletBatch

        gneg = torch.Generator().manual_seed(99 + rank)
        negs = [((torch.arange(B) + 1 + torch.randint(0, B - 1, (B,), generator=gneg)) % B).int().to(device)
                for _ in range(n_distinct)]
        if args.inputs == "index":
            batches = [TripletBatch(ids=ids[:2].contiguous(), neg_index=ni) for ids, ni in zip(id_batches, negs)]
        else:
            n_texts = 400_000  # e.g. 200k queries + 200k passages, pooled once: 400k x D x 4 B
            cache_ids = torch.randint(0, V, (n_texts, S), generator=torch.Generator().manual_seed(5), dtype=torch.int32)
            trainer.build_feature_cache(cache_ids.to(device))
            del cache_ids
            grow = torch.Generator().manual_seed(7 + rank)
            batches = [TripletBatch(cache_rows=torch.randint(0, n_texts, (2, B), generator=grow, dtype=torch.int32).to(device),
                                    neg_index=ni) for ni in negs]
    eng = trainer.engine

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    k1_ev = []
    depth = trainer.SLOTS - 1 if trainer.side is not None else 1
    issued = [0]

    if args.h2d:
        from two_towers_overlords_amd.training import TripletBatch as _TB

        host_ids = [ids.cpu().pin_memory() for ids in id_batches]
        copy_stream = torch.cuda.Stream(device)
        copy_events = [torch.cuda.Event() for _ in range(4)]

    def next_batch():
        k = issued[0]
        issued[0] += 1
        if args.h2d:  # host -> device on the copy stream; the gather waits for the copy's event
            with torch.cuda.stream(copy_stream):
                ids = host_ids[k % n_distinct].to(device, non_blocking=True)
                ev = copy_events[k % 4]
                ev.record(copy_stream)
            return _TB(ids=ids, ready_event=ev)
        return batches[k % n_distinct]

    def fill():
        while len(trainer._pending) < depth:
            trainer.gather(next_batch())

    def run(n_steps, timed):
        """n_steps steady-state steps of TripletTrainer.train_epoch's loop: the oldest gathered batch goes through
        fwd + bwd + Adam, the gather of the batch `depth` steps ahead is enqueued."""
        trainer.k1_events = k1_ev if timed else None
        for k in range(1, n_steps + 1):
            if args.k1_only:  # gathers only: drop the oldest pooled batch unprocessed
                trainer._pending.popleft()
                trainer.gather(next_batch())
            else:
                trainer.advance(next_batch(), B * world)
            if timed and trace_ev and k % trace_every == 0 and len(trace_host) < len(trace_ev):
                trace_ev[len(trace_host)].record()
                trace_host.append(time.perf_counter())
        trainer.k1_events = None

    def drain():
        while trainer._pending:
            if args.k1_only:
                trainer._pending.popleft()
            else:
                trainer.advance(None, B * world)

    trace_every = 1 if args.debug_trace == "step" else 10  # --debug-trace step: one event per step
    trace_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps // trace_every)] if args.debug_trace else []
    trace_host = []

    if not args.debug_no_gc_freeze:  # (--debug-no-gc-freeze: A/B of this very effect)
        # A full (generation-2) pass of Python's cyclic GC over the ~10^6 objects that importing torch leaves
        # behind takes 60-90 ms of host time; it used to land inside about one timed region in fifteen and
        # starve the GPU (step 200 -> 500 us).  Freeze what exists now: later passes only look at new objects.
        import gc
        gc.collect()
        gc.freeze()
    if args.debug_prewarm_ms:  # experiment: keep the GPU busy for a while before warm-up
        a = torch.randn(4096, 4096, device=device)
        tend = time.perf_counter() + args.debug_prewarm_ms * 1e-3
        while time.perf_counter() < tend:
            a @ a
            torch.cuda.synchronize(device)
        del a
    # (every piece of host-side set-up goes BEFORE calibrate(): its dry steps bring the GPU to its sustained power state, and
    #  a millisecond of idleness afterwards already costs the next ~30 steps a few percent — docs/LOG_r04.md section 5, "Cold start")
    trainer.k1_event_pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                             for _ in range(args.steps // max(trainer.k1_event_every, 1) + 8)]
    for e0, e1 in trainer.k1_event_pool:  # torch creates the HIP event at its first record(): do that now
        e0.record()
        e1.record()
    if not args.no_calibrate and not args.k1_only and args.loss == "triplet":
        trainer.calibrate(batches[0], steps=getattr(args, "calibrate_steps", 48),
                          rounds=getattr(args, "calibrate_rounds", 2),   # set-up: part of what run_training does before its first epoch
                          k1_forms=(None if getattr(args, "calibrate_k1", None) is None else bool(args.calibrate_k1)))
    set_up_steps = 0  # steps of this workload executed before the timed region besides --warmup
    if trainer.calibration:
        combos = len(trainer.calibration.get("ms_per_step_by_fence") or (0, 0))  # backward schedules x fence placements timed
        set_up_steps += combos * trainer.calibration["rounds"] * (trainer.calibration["steps"] + depth)  # dry fwd+bwd (no optimizer)
    dp_breakdown = None
    if args.e2e:
        elapsed, e2e_info = run_e2e(args, trainer, cfg, device, rank, world, barrier)
        host_issue = elapsed
        k1_ev = e2e_info["k1_events"]
    else:
        fill()
        if not args.no_calibrate:  # set-up, like calibrate(): run the event-timed code path once before anything is timed
            run(4, timed=True)
            k1_ev.clear()
            trace_host.clear()
            set_up_steps += 4
        run(args.warmup, timed=False)
        barrier()
        if args.debug_idle_ms:  # experiment: idle gap between warm-up and the timed region
            time.sleep(args.debug_idle_ms * 1e-3)
        t0 = time.perf_counter()
        run(args.steps, timed=True)
        host_issue = time.perf_counter() - t0  # all K steps enqueued (no device sync inside the loop)
        barrier()
        elapsed = time.perf_counter() - t0
        if not args.k1_only and args.trace_steps > 1 and args.loss == "triplet" and not trainer.use_graphs:
            # a separate short pass: every step carries four timing events on the caller's stream (each a packet on the
            # critical chain), so its steps are a few us longer than the timed region's — reported beside them
            trainer.step_trace = []
            run(args.trace_steps, timed=False)
            torch.cuda.synchronize(device)
            tr_ev, trainer.step_trace = trainer.step_trace, None
            dp_breakdown = breakdown(tr_ev)
        drain()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    loss_avg = e2e_info["loss"] if args.e2e else trainer.epoch_loss()

    # K1 alone (nothing else on the chip), same engine / residency cap, after the timed region
    iso_ev = []
    n_iso = getattr(args, "iso_launches", 20)
    iso_roles = 3 if args.inputs == "ids3" else 2  # what a step of this mode gathers (index / in-batch modes: q and pos only)
    for i in range(n_iso):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        bb = batches[i % n_distinct] if (args.ragged and args.inputs == "ids3") else (id_batches[i % n_distinct], None, None)
        e0.record(); trainer.engine.pool(i & 1, bb[0][:iso_roles].contiguous(), None, bb[2]); e1.record()
        iso_ev.append((e0, e1))
    torch.cuda.synchronize(device)
    iso_ms = sum(a.elapsed_time(b) for a, b in iso_ev[n_iso // 5:]) / len(iso_ev[n_iso // 5:])
    k1_ms = sum(a.elapsed_time(b) for a, b in k1_ev) / max(len(k1_ev), 1)
    roles = 3 if args.inputs == "ids3" else 2
    alg = k1_algorithmic_bytes(roles * B, S, D)
    if args.ragged and args.inputs == "ids3":  # n_valid*D*4 + S*4 + D*4 per text with the actual lengths
        alg = int(valid_tokens * D * 4 + roles * B * (S * 4 + D * 4))
    from two_towers_overlords_amd import _lib as _ttlib
    sliced = bool(_ttlib.load().tt_k1_prefers_sliced(V, D, roles * B, S))
    streaming = (trainer.side is not None and trainer.engine_k1_cap < 0
                 and bool(_ttlib.load().tt_k1_prefers_stream(V, D, roles * B, S)))
    columns = sliced and _ttlib.load().tt_k1_sliced_form(V, D, S) == _ttlib.TT_K1_COLUMN_SLICES
    binned = bool(_ttlib.load().tt_k1_prefers_binned(V, D, roles * B, S))  # the engine's rule: binned before streaming
    streaming = streaming and not binned
    forced = (trainer.engine.tuning or {}).get("k1_form") if hasattr(trainer.engine, "tuning") else None
    if forced:  # --tune k1_form=...: the engine was told which form to use
        streaming, sliced = forced == _ttlib.TT_K1_STREAM, forced in (_ttlib.TT_K1_ROW_SLICES, _ttlib.TT_K1_COLUMN_SLICES)
        columns, binned = forced == _ttlib.TT_K1_COLUMN_SLICES, forced == _ttlib.TT_K1_BINNED
    k1_kernel = ("pool_binned_kernel" if binned else "pool_colsliced_kernel" if columns else "pool_sliced_kernel" if sliced else
                 "pool_norm_stream_kernel" if (streaming or args.k1_only and forced == _ttlib.TT_K1_STREAM) else "pool_norm_kernel")
    traffic = traffic_source = None  # fabric-side bytes per K1 launch: from the committed PMC passes of this same workload AND kernel, if any
    import glob
    tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_k1_traffic_{args.config}.json")))
    if tpaths and not args.zipf and not args.ragged and args.inputs == "ids3":  # the newest round's PMC passes
        tj = json.load(open(tpaths[-1]))
        if int(tj.get("algorithmic_bytes_per_launch", -1)) == alg and tj.get("kernel", "").split(",")[0].split("*")[0] == k1_kernel:
            traffic = tj["traffic_bytes_per_launch"]
            traffic_source = (os.path.relpath(tpaths[-1], ROOT) + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                              "workload and kernel (`bench.py --k1-only`), committed; PMC counters are not collected in a timed run")
    achieved = alg / (k1_ms * 1e-3) / 1e9
    k1_label = ("pool_bin_kernel + pool_binned_kernel + pool_bincombine_kernel (K1, binned form: each text's tokens sorted "
                "by row class, then class x 32-column-slice blocks of the table gathered out of the XCDs' L2s, the 8 class "
                "partials added, mean, L2 norm; the three launches lie inside the timed events)" if binned else
                "pool_colsliced_kernel + pool_colnorm_kernel (K1, XCD-sliced form, 32-column slices — one 128-byte line per "
                "row, a slice fits the XCD's L2: gather + masked mean-pool, then the L2 norm; both launches lie inside the "
                "timed events)" if columns
                else "pool_sliced_kernel + pool_combine_kernel (K1, XCD-sliced form, row slices: gather + masked mean-pool + "
                     "L2 normalise; both launches lie inside the timed events)" if sliced
                else "pool_norm_stream_kernel (K1, streaming form: persistent workgroups, two per compute unit, pulling texts "
                     "from a queue; gather + masked mean-pool + L2 normalise)" if streaming
                else "pool_norm_kernel (K1 gather + masked mean-pool + L2 normalise)")
    metric = (("triplets/sec (q,pos,neg @ seq128/d384), full training step" if args.config == "c2"
               else "triplets/sec, full training step") if args.inputs == "ids3"
              else f"triplets/sec, full training step, inputs={args.inputs}, loss={args.loss} (NOT the BASELINE metric)")
    if args.simulate_allreduce_us is not None:
        metric = (f"triplets/sec, data-parallel step on ONE GPU with the all-reduce replaced by a {args.simulate_allreduce_us:g} us "
                  "stand-in kernel (diagnostic; NOT the BASELINE metric)")
    if args.h2d:
        metric = "triplets/sec, full training step, ids copied from pinned host memory every step (PCIe-inclusive; NOT the BASELINE metric)"
    if args.e2e:
        metric = (f"triplets/sec, one whole epoch of {args.steps} batches through TokenTripletLoader(sampler=device) + "
                  f"TripletTrainer.train_epoch, inputs={args.inputs} (end-to-end loop; NOT the BASELINE metric)")
    # `bound` / `peak` / `frac` are the contract's figure on SURVEY.md section 8(d)'s definition: ALGORITHMIC bytes per launch ÷
    # the launch's duration against the nominal 8 TB/s of HBM — the same denominator in every round (rounds 1-4 quoted it
    # as `frac`, round 5 as `nominal_hbm.frac`), so the trend stays readable.  Against a cache-resident table that figure is
    # not a physical bound (rows named many times are served by the caches; it can exceed 1), so what actually bounds the
    # launch is spelled out beside it: `binding` says where the rows come from, `l2_peak` is the XCD L2s' aggregate rate
    # (MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s), `reference_loop_rate` the guide's MEASURED rate of one LDS-staging
    # gather loop over L2-resident shared rows (16.8-18.8 TB/s: "Per-CU rates (lower bounds)", not a hardware peak).
    cache_resident = V * D * 4 <= 200e6
    l2_bound = cache_resident or binned
    achieved_alone = alg / (iso_ms * 1e-3) / 1e9
    table_bytes = V * D * 4
    crossings = (traffic / table_bytes) if (traffic and binned) else None
    roofline = {
        "kernel": k1_label, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "peak_source": "MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)",
        "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": k1_ms,
        "launches_timed": len(k1_ev),
        "binding": "l2" if l2_bound else "hbm",
        "bound_note": ("the table (%.0f MB) is cache-resident: rows shared by every workgroup come out of the XCDs' L2s and the "
                       "Infinity Cache, so the nominal-HBM `frac` is a rate of algorithmic bytes, not a fraction of a physical "
                       "bound (alone it exceeds 1); `l2_peak` / `reference_loop_rate` are the rates that bind" % (table_bytes / 1e6)) if cache_resident else
                      (("the table (%.0f MB) exceeds the 256 MiB Infinity Cache; the binned form gathers it as (row class x 32-column "
                        "slice) blocks out of the XCDs' L2s.  " % (table_bytes / 1e6))
                       + (("Committed counters: %.2f GB cross the L2s' fabric side per launch = %.1f x the table (a 4.0 MB block "
                           "shares its 4 MiB L2 with the id and partial-sum streams, so lines are re-fetched), against %.2f GB of "
                           "algorithmic bytes; launch time is insensitive to the block size (profiles/r05_binned_k1.txt), so the L2s' "
                           "delivery rate, not the fabric, binds" % (traffic / 1e9, crossings, alg / 1e9)) if crossings else
                          "No committed counter pass matches this workload, so nothing is claimed about fabric traffic.")) if binned else
                      "the table (%.0f MB) exceeds the 256 MiB Infinity Cache and streams from HBM" % (table_bytes / 1e6),
        "table_fabric_crossings_per_launch": crossings,
        "l2_peak": {"peak": L2_PEAK_GBS, "unit": "GB/s", "frac": achieved / L2_PEAK_GBS, "alone_frac": achieved_alone / L2_PEAK_GBS,
                    "source": "MI355X_MICROARCH.md 'L2 (per XCD)': 4 MiB per XCD, ~34.5 TB/s aggregate", "applies": l2_bound},
        "reference_loop_rate": {"range": list(L2_SHARED_ROWS_GBS), "unit": "GB/s", "frac_of_upper": achieved / L2_SHARED_ROWS_GBS[1],
                                "alone_frac_of_upper": achieved_alone / L2_SHARED_ROWS_GBS[1], "applies": l2_bound,
                                "source": "MI355X_MICROARCH.md 'Indexed rows: gather into LDS': 66-73 GB/s per CU x 256 CUs — the "
                                          "measured rate of ONE staging loop over 2,048 L2-resident shared rows (the guide labels the "
                                          "table 'lower bounds'); round 5 quoted its upper figure as `peak`"},
        "note": "timed inside the step loop (HIP events on the gather's stream); with gather_overlap the launch shares the "
                "chip with the projection kernels of the previous batch, so it takes longer than alone.  `achieved` counts "
                "algorithmic bytes (every row read, SURVEY.md section 8d); `traffic` is what crossed the L2s' fabric side",
        "served_from": ("XCD L2 + Infinity Cache (the table fits the 256 MiB Infinity Cache)" if cache_resident else
                        "XCD L2s, block by block (binned form; the table exceeds the Infinity Cache)" if binned
                        else "HBM (the table exceeds the Infinity Cache)"),
        "alone": {"avg_launch_ms": iso_ms, "achieved": achieved_alone, "frac": achieved_alone / HBM_PEAK_GBS},
        "nominal_hbm": {"peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                        "alone_frac": achieved_alone / HBM_PEAK_GBS,
                        "is_the_binding_bound": not l2_bound},
        "k2": (None if args.vocab else k2_mfma_util(f"in-batch negatives B={B} ") if args.loss == "inbatch"
               else None if args.batch else k2_mfma_util(args.config)),
    }
    result = {
        "metric": metric,
        "value": args.steps * B * world / elapsed,
        "unit": "triplets/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "effective_warmup_steps": args.warmup + set_up_steps,  # + calibrate()'s dry fwd+bwd steps + the 4 event-timed set-up steps
        "ms_per_step": elapsed / args.steps * 1e3,
        "host_issue_ms_per_step": host_issue / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["label"], "encoder_mode": "embed", "vocab": V, "embed_dim": D, "proj_dim": P,
                   "seq_len": S, "batch_per_gpu": B, "global_batch": B * world,
                   "rccl_ranks": (dist.get_world_size() if (world > 1 and args.backend == "nccl") else (1 if world == 1 else 0)),
                   "backend": (args.backend if world > 1 else None), "ids": "int32 [3,B,S], "
                   + ("zipf" if args.zipf else "uniform") + (", ragged lengths" if args.ragged else ", full length"), "parallelism": f"dp{world}", "inputs": args.inputs,
                   "k1_only": bool(args.k1_only), "tuning": (trainer.engine.tuning or None),
                   "event_scope": getattr(trainer, "event_scope", None), "grad_slabs": trainer.engine.n_slabs,
                   "trainer": (getattr(args, "trainer_cfg", None) or None),
                   "gather_overlap": ("side stream, " + ("K1's LDS reservation paired with the projection tile (auto)"
                                                       if trainer.engine_k1_cap < 0 else
                                                       "K1 capped at %d workgroups/CU" % trainer.engine_k1_cap))
                   if trainer.side is not None else "none (same stream)",
                   "timed_region": ("one epoch: loader + pipeline fill and drain + loss read-back" if args.e2e else
                                    "K steady-state steps (pipeline filled in warm-up, drained after)"),
                   "launch": "hipGraph replay (1 step per graph; every 8th step eager for K1 event timing)"
                   if trainer.use_graphs else "eager (kernel by kernel)"},
        "roofline": roofline,
        "avg_loss": loss_avg,
        "calibration": trainer.calibration,
    }
    if dp_breakdown is not None:
        dp_breakdown.update({
            "timed_ms_per_step": elapsed / args.steps * 1e3,
            "k1_in_loop_ms": k1_ms, "k1_alone_ms": iso_ms,
            "k1_placement": ("under the exchange (issued right before it, side stream)" if trainer._exchange and trainer.side is not None
                             else "beside the chain (side stream)" if trainer.side is not None else "same stream, ahead of the chain"),
            "exchange": (None if not trainer._exchange else
                         {"form": "two-part" if trainer._split else "one-bucket", "bucket_bytes": int(trainer.bucket.numel() * 4),
                          "backend": (args.backend if world > 1 else "stand-in kernel" if sim else "none"),
                          "rccl_ranks": (dist.get_world_size() if (world > 1 and args.backend == "nccl") else 0)}),
            "note": "device events on the caller's stream, a separate traced pass of --trace-steps steps; chain + exchange + "
                    "adam + other = traced_ms_per_step; the traced steps carry four event packets each, so they run a few "
                    "us longer than the timed region's (timed_ms_per_step)"})
        if world > 1:  # every rank's numbers travel to rank 0: the slowest chain / longest exchange is what the step waits for
            allb = [None] * world
            dist.all_gather_object(allb, {k: dp_breakdown[k] for k in ("chain_ms", "exchange_ms", "adam_ms", "other_ms", "traced_ms_per_step", "k1_in_loop_ms")})
            dp_breakdown["per_rank"] = allb
        result["dp_breakdown"] = dp_breakdown
    if args.inputs == "cache":  # no gather in the step: the K1 roofline does not apply
        result["roofline"] = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not getattr(args, "cpu_baseline_later", False):
        result["cpu_baseline"] = cpu_baseline(cfg)
    if trace_ev and len(trace_host) > 1:
        gaps = [round(trace_ev[i].elapsed_time(trace_ev[i + 1]), 2) for i in range(len(trace_host) - 1)]
        hgaps = [round((trace_host[i + 1] - trace_host[i]) * 1e3, 2) for i in range(len(trace_host) - 1)]
        print(f"TRACE gpu ms per {trace_every} step(s):", gaps, file=sys.stderr)
        print(f"TRACE host ms per {trace_every} step(s):", hgaps, file=sys.stderr)
        print("TRACE K1 in-loop ms (every %d-th launch):" % trainer.k1_event_every,
              [round(a.elapsed_time(b), 3) for a, b in k1_ev], file=sys.stderr)
    if args.debug_trace and hasattr(trainer.engine, "k1_census"):
        print("TRACE streaming-K1 census per slot (admitted workgroups, CUs used, max candidates on one CU, planned):",
              [trainer.engine.k1_census(s) for s in range(trainer.SLOTS)], file=sys.stderr)
    trainer.engine.close()  # the workspace goes back to the allocator before a `secondary` configuration is set up
    return result


if __name__ == "__main__":
    main()
