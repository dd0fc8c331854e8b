#!/bin/bash
# Round 3: split-bf16 MFMA products (TT_BF3=1: split on the fragments; TT_BF3=2: split once per workgroup at staging
# time; 3: that for kNT only) — parity under the flag, then A/B per workload.   -> gpurun_out/bf3/
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/bf3; mkdir -p $out
cd $root
python3 -m pytest tests/test_gpu_stream_policy.py -q -m gpu -k "BF3" > $out/pytest.txt 2>&1; rc=$?
tail -4 $out/pytest.txt | cut -c1-250
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"] or {}; print(round(j["ms_per_step"]*1e3,1), "us  loss", j["avg_loss"], " K1 in-loop", round((r.get("avg_launch_ms") or 0)*1e3,1))'
rm -f $out/ab2.txt
for rep in 1 2; do for bf in 0 3 2; do
echo "c2 default TT_BF3=$bf: $(TT_BF3=$bf python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/ab2.txt
echo "c2 cache TT_BF3=$bf: $(TT_BF3=$bf python3 bench.py --no-cpu-baseline --no-secondary --inputs cache | python3 -c "$pick")" >> $out/ab2.txt
echo "inbatch b4096 TT_BF3=$bf: $(TT_BF3=$bf python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096 | python3 -c "$pick")" >> $out/ab2.txt
echo "inbatch b4096 one stream TT_BF3=$bf: $(TT_BF3=$bf python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096 --no-overlap | python3 -c "$pick")" >> $out/ab2.txt
done; done
for bf in 0 3 2; do
echo "c4 one stream TT_BF3=$bf: $(TT_BF3=$bf python3 bench.py --no-cpu-baseline --config c4 --steps 16 --warmup 4 --no-overlap | python3 -c "$pick")" >> $out/ab2.txt
echo "c4 TT_BF3=$bf: $(TT_BF3=$bf python3 bench.py --no-cpu-baseline --config c4 --steps 30 --warmup 8 | python3 -c "$pick")" >> $out/ab2.txt
done
sort $out/ab2.txt
exit $rc
