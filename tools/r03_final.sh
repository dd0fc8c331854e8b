#!/bin/bash
# split-bf16 parity under the flag, then the round's profile refresh
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/final; mkdir -p $out
cd $root
python3 -m pytest tests/test_gpu_stream_policy.py -q -m gpu -k "BF3" > $out/pytest_bf3.txt 2>&1; rc=$?
tail -4 $out/pytest_bf3.txt | cut -c1-300
bash tools/refresh_profiles.sh r03 > $out/refresh.log 2>&1; rc2=$?
tail -25 $out/refresh.log
exit $((rc + rc2))
