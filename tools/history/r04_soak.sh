#!/bin/bash
O=gpurun_out/r04_soak; mkdir -p $O
timeout -k 10 300 python tools/determinism_probe.py 4000 triplet > $O/determinism_triplet.txt 2>&1; echo "rc=$?" >> $O/determinism_triplet.txt; tail -4 $O/determinism_triplet.txt
timeout -k 10 300 python tools/determinism_probe.py 1600 inbatch > $O/determinism_inbatch.txt 2>&1; echo "rc=$?" >> $O/determinism_inbatch.txt; tail -4 $O/determinism_inbatch.txt
timeout -k 10 300 python tools/soak.py 40 > $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt; tail -3 $O/soak.txt
timeout -k 10 200 python tools/lifecycle.py > $O/lifecycle.txt 2>&1; echo "rc=$?" >> $O/lifecycle.txt; tail -3 $O/lifecycle.txt
