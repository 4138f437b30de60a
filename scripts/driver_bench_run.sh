#!/bin/bash
# The driver's bench command, timed, + smoke()
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/driver_bench; rm -rf $OUT; mkdir -p $OUT
t0=$(date +%s)
timeout -k 10 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; rc=$?
t1=$(date +%s); echo "bench rc=$rc wall=$((t1-t0))s"
tail -4 $OUT/bench.err
if [ $rc -ne 0 ]; then exit 1; fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
