#!/bin/bash
# round 3, GPU call B: parity of the band-ordered role-fused launches, sweep of the band plan
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3b; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_bands.py -x -q -m gpu > $OUT/bands.log 2>&1; rc=$?; echo "bands rc=$rc" | tee -a $OUT/summary.txt
tail -15 $OUT/bands.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/parity.log 2>&1; rc=$?; echo "parity rc=$rc" | tee -a $OUT/summary.txt
tail -8 $OUT/parity.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python scripts/transform_sweep.py $OUT/sweep.json > $OUT/sweep.log 2>&1; echo "sweep rc=$?" | tee -a $OUT/summary.txt
tail -50 $OUT/sweep.log
