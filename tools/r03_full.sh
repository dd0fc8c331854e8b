#!/bin/bash
# GPU suite + default bench line (with `secondary`) + gemm16 A/B at configs[1] + in-batch timing
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/full; mkdir -p $out
cd $root
python3 -m pytest tests -q -m gpu ${PYTEST_K:+-k "$PYTEST_K"} > $out/pytest_gpu.txt 2>&1; rc=$?
tail -5 $out/pytest_gpu.txt
[ $rc = 0 ] || exit 1
( time python3 bench.py > $out/bench_default.json ) 2> $out/bench_default.time
cat $out/bench_default.json | python3 -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j["value"], j["ms_per_step"], j["roofline"]["frac"], "sec:", j["secondary"]["value"], j["secondary"]["ms_per_step"], j["secondary"]["roofline"]["frac"], j["secondary"]["roofline"]["alone"]["frac"], "bert:", j["cpu_baseline_bert"]["value"], "embed:", j["cpu_baseline"]["value"])'
tail -3 $out/bench_default.time
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"]*1e3,1), "us")'
rm -f $out/g16.txt
for rep in 1 2; do for g in 0 1; do
  echo "cache TT_G16=$g: $(TT_G16=$g python3 bench.py --no-cpu-baseline --no-secondary --inputs cache | python3 -c "$pick")" >> $out/g16.txt
  echo "default TT_G16=$g: $(TT_G16=$g python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/g16.txt
done; done
echo "cache auto: $(python3 bench.py --no-cpu-baseline --no-secondary --inputs cache | python3 -c "$pick")" >> $out/g16.txt
echo "no-overlap auto: $(python3 bench.py --no-cpu-baseline --no-secondary --no-overlap | python3 -c "$pick")" >> $out/g16.txt
echo "no-overlap TT_G16=0: $(TT_G16=0 python3 bench.py --no-cpu-baseline --no-secondary --no-overlap | python3 -c "$pick")" >> $out/g16.txt
echo "inbatch b4096: $(python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096 | python3 -c "$pick")" >> $out/g16.txt
echo "inbatch b2048: $(python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch | python3 -c "$pick")" >> $out/g16.txt
echo "index b16384 c4: $(python3 bench.py --no-cpu-baseline --config c4 --inputs index --steps 12 --warmup 4 | python3 -c "$pick")" >> $out/g16.txt
cat $out/g16.txt
