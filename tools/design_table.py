#!/usr/bin/env python3
"""Regenerates the table of DESIGN.md section 5 from profiles/rNN_bench_*.json (between the markers
<!-- bench-table-begin --> and <!-- bench-table-end -->), so the table is exactly what the committed JSON lines say.
    usage: tools/design_table.py [rNN]   (default r06)"""
import json
import os
import re
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r06"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def j(name):
    p = os.path.join(P, name)
    return json.load(open(p)) if os.path.exists(p) else None


def row(label, name, unit="M"):
    r = j(name)
    if r is None:
        return None
    rf = r.get("roofline") or {}
    nom = (rf.get("nominal_hbm") or {}).get("frac")
    l2 = rf.get("l2_peak") or {}
    k1 = (f"{rf['achieved'] / 1e3:.2f} TB/s = {nom if nom is not None else rf['frac']:.2f} of 8 TB/s HBM"
          + (f"; {l2['frac']:.2f} of the L2s' 34.5 TB/s" if l2.get("applies") else "")) if rf.get("achieved") else "—"
    alone = f"{rf['alone']['achieved'] / 1e3:.2f} TB/s = {rf['alone']['avg_launch_ms'] * 1e3:.1f} µs" if rf.get("alone") and rf.get("achieved") else "—"
    step = r["ms_per_step"] * 1e3
    step_s = f"{step / 1e3:.2f} ms" if step >= 2000 else f"{step:.1f} µs"
    return f"| {label} `{name}` | {r['value'] / 1e6:.2f} M | {step_s} | {k1} | {alone} |"


rows = [
    ("**configs[1] default** (the driver's line: carries `secondary`, three CPU baselines, `dp_breakdown`)", R + "_bench_c2_default.json"),
    ("configs[1] at the driver's 20 steps / 5 warm-up", R + "_bench_c2_default_k20_w5.json"),
    ("whole epoch through the loader, 1000 batches (`--e2e`)", R + "_bench_c2_e2e_epoch1000.json"),
    ("`--no-overlap` (everything on one stream)", R + "_bench_c2_no_overlap.json"),
    ("`--ragged` lengths", R + "_bench_c2_ragged.json"),
    ("`--zipf` ids (SURVEY §8d's text-like id distribution)", R + "_bench_c2_zipf.json"),
    ("`--inputs index` (negatives by in-batch index, 2 gathers)", R + "_bench_c2_inputs_index.json"),
    ("`--inputs cache` (cache-fed step: no gather)", R + "_bench_c2_inputs_cache.json"),
    ("**`--path module`**: the reference's literal call `train_epoch(model, batches, TripletLoss(), torch.optim.Adam(model.parameters(), lr))` on pre-tokenised int64 (ids, mask) per role — batch body routed to the engine; fill, drain, engine-side role packing and the loss reads inside the timed epoch", R + "_bench_c2_module_path.json"),
    ("… with `--optimizer fused` (`FusedAdam`)", R + "_bench_c2_module_path_fused_adam.json"),
    ("`--path module-unfused`: the same call with `fused=False` (autograd Functions, module by module, torch's own Adam kernels)", R + "_bench_c2_module_unfused.json"),
    ("**`--path module --strings columns`**: STRING batches (real `BertTokenizerFast`) as the row-tagged lists this package's `TripletDataLoader` yields; best epoch ≥ 2 (text store, no tokenizer call)", R + "_bench_c2_strings_columns.json"),
    ("`--strings lists`: plain `list[str]` batches, best epoch ≥ 2 (one dict look-up per text on the host)", R + "_bench_c2_strings_lists.json"),
    ("`--strings loader`: through `TripletDataLoader` itself (the reference's shuffle + negative draws per batch), best epoch ≥ 2", R + "_bench_c2_strings_loader.json"),
    ("`--strings lists --text-cache 0`: three tokenizer calls per batch, every epoch — what the reference does", R + "_bench_c2_strings_lists_nocache.json"),
    ("configs[2], one GPU's share (B = 1024 of 8192, P = 512)", R + "_bench_c3_share.json"),
    ("**configs[3]** (V = 250k, D = P = 768, S = 256, B = 16384; binned K1)", R + "_bench_c4_full.json"),
    ("configs[3] with the streaming K1 (`--tune k1_form=stream`: round 3's form, same box)", R + "_bench_c4_stream_form.json"),
    ("configs[3] `--zipf` ids, DEFAULT (round 6: `calibrate()` times both gather forms for a table beyond the Infinity Cache and picked the streaming one)", R + "_bench_c4_zipf.json"),
    ("configs[3] `--zipf` ids, streaming K1 forced (`--tune k1_form=stream`)", R + "_bench_c4_zipf_stream_form.json"),
    ("configs[3] `--ragged` lengths, binned K1", R + "_bench_c4_ragged.json"),
    ("configs[3] `--ragged` lengths, streaming K1", R + "_bench_c4_ragged_stream_form.json"),
    ("configs[3] one stream `--no-overlap`", R + "_bench_c4_no_overlap.json"),
    ("configs[3] negatives by index", R + "_bench_c4_inputs_index.json"),
    ("in-batch negatives B = 4096 (queries/s)", R + "_bench_inbatch_b4096.json"),
    ("in-batch negatives B = 2048", R + "_bench_c2_inbatch_b2048.json"),
    ("in-batch negatives, one GPU's share of configs[4] (B = 512 of 4096)", R + "_bench_inbatch_b512_share.json"),
]
out = ["| run (`profiles/…`) | triplets/s | per step | K1 in the loop (algorithmic bytes ÷ launch time; against the nominal 8 TB/s of HBM, and against the L2s where they bind) | K1 alone |", "|---|---|---|---|---|"]
for label, name in rows:
    r = row(label, name)
    if r:
        out.append(r)
d = j(R + "_bench_c2_default.json")
if d:
    s = d["secondary"]
    out.append(f"| configs[3] as `secondary` of the default line | {s['value'] / 1e6:.2f} M | {s['ms_per_step']:.2f} ms | "
               f"{s['roofline']['achieved'] / 1e3:.2f} TB/s = {s['roofline']['frac']:.2f} of {s['roofline'].get('bound', 'hbm')} | {s['roofline']['alone']['achieved'] / 1e3:.2f} TB/s |")
    for k, lab in (("cpu_baseline", "CPU oracle `embed` mode at configs[1]'s shape"), ("cpu_baseline_configs0", "CPU oracle at configs[0]'s shape (B = 256, P = 64)"),
                   ("cpu_baseline_bert", "CPU oracle `bert` mode — what the reference's CPU path executes")):
        if d.get(k) and d[k].get("value"):
            out.append(f"| {lab} (`{k}`, {d[k]['cores']} host threads) | {d[k]['value']:,.0f} /s | | | |")
table = "\n".join(out)
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
s2 = re.sub(r"<!-- bench-table-begin -->.*?<!-- bench-table-end -->", "<!-- bench-table-begin -->\n" + table + "\n<!-- bench-table-end -->", s, flags=re.S)
assert s2 != s or table in s, "markers not found"
open(p, "w").write(s2)
print(table)
