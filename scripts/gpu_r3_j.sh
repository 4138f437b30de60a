#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3j; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stage_semantics.py tests/test_gpu_bands.py tests/test_cli.py tests/test_cpp_mirror.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -8 $OUT/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python scripts/host_api_probe.py $OUT/host_api.json 1 8 64 > $OUT/host_api.log 2>&1; echo "host api rc=$?"
grep "host api" $OUT/host_api.log
