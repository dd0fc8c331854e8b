#!/bin/bash
# whole GPU suite, one process
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/suite; mkdir -p $out
cd $root
python3 -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1; rc=$?
tail -15 $out/pytest_gpu.txt | cut -c1-300
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
exit $rc
