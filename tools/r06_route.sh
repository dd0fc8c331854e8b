#!/bin/bash
# Round 6: the reference's literal call on the engine — stock torch.optim.Adam, string batches.
# usage (on the GPU box): bash tools/r06_route.sh [out_dir]
set -o pipefail
out=${1:-gpurun_out/r06_route}
mkdir -p "$out"
b() { name=$1; shift; echo "== $name: bench.py $*"; timeout -k 10 400 python bench.py "$@" > "$out/$name.json" 2> "$out/$name.log" || { echo "FAILED $name"; tail -5 "$out/$name.log"; return 1; }; python -c "
import json,sys
r=json.load(open('$out/$name.json'))
print('   value %.3f M/s  %.1f us/step  routed=%s' % (r['value']/1e6, r['ms_per_step']*1e3, r.get('routed')))
for e in r.get('epochs', []): print('   epoch %d: %.3f M/s, %.0f us/batch, tokenizer share %s' % (e['epoch'], e['triplets_per_s']/1e6, e['us_per_batch'], e['tokenizer_share']))
"; }
b module_torch_adam        --path module --optimizer torch --steps 400 --warmup 50 &&
b module_fused_adam        --path module --optimizer fused --steps 400 --warmup 50 &&
b module_unfused_torch_adam --path module-unfused --optimizer torch --steps 100 --warmup 20 &&
b strings_columns          --path module --strings columns --steps 64 --epochs 12 &&
b strings_lists            --path module --strings lists --steps 64 --epochs 12 &&
b strings_loader           --path module --strings loader --steps 16 --epochs 6 &&
b strings_lists_nocache    --path module --strings lists --steps 4 --epochs 2 --text-cache 0
