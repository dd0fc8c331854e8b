#!/bin/bash
# Round 3: split-bf16 MFMA products (TT_BF3=1; opt-in experiment) — parity under the flag, then A/B per workload.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/bf3; mkdir -p $out
cd $root
python3 -m pytest tests/test_gpu_stream_policy.py -q -m gpu -k "BF3" > $out/pytest.txt 2>&1; rc=$?
tail -25 $out/pytest.txt | cut -c1-250
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"] or {}; print(round(j["ms_per_step"]*1e3,1), "us  loss", j["avg_loss"], " K1 in-loop", round((r.get("avg_launch_ms") or 0)*1e3,1))'
rm -f $out/ab.txt
for bf in 0 1; do
echo "c2 default TT_BF3=$bf: $(TT_BF3=$bf python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab.txt
echo "c2 cache TT_BF3=$bf: $(TT_BF3=$bf python3 bench.py --no-cpu-baseline --no-secondary --inputs cache | python3 -c "$pick")" >> $out/ab.txt
echo "c2 no-overlap TT_BF3=$bf: $(TT_BF3=$bf python3 bench.py --no-cpu-baseline --no-secondary --no-overlap | python3 -c "$pick")" >> $out/ab.txt
echo "inbatch b4096 TT_BF3=$bf: $(TT_BF3=$bf python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096 | python3 -c "$pick")" >> $out/ab.txt
echo "c4 TT_BF3=$bf: $(TT_BF3=$bf python3 bench.py --no-cpu-baseline --config c4 --steps 30 --warmup 8 | python3 -c "$pick")" >> $out/ab.txt
echo "c4 no-overlap TT_BF3=$bf: $(TT_BF3=$bf python3 bench.py --no-cpu-baseline --config c4 --steps 16 --warmup 4 --no-overlap | python3 -c "$pick")" >> $out/ab.txt
done
cat $out/ab.txt
exit $rc
