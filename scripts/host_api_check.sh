#!/bin/bash
# Chain hub: parity of the host paths (incl. concurrency tests), then the host-api timing with T threads
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/host_api_check; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stage_semantics.py tests/test_gpu_multi.py tests/test_gpu_value_table.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $OUT/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python scripts/host_api_probe.py $OUT/host_api.json ${@:-1 8 64} > $OUT/host_api.log 2>&1; echo "probe rc=$?"
grep "host api" $OUT/host_api.log; tail -3 $OUT/host_api.log | cut -c1-600
