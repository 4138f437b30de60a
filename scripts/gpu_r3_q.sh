#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3q; rm -rf $OUT; mkdir -p $OUT
ALICE_CODEC_DEBUG=1 timeout -k 10 400 python scripts/host_api_probe.py $OUT/host_api.json 64 > $OUT/host_api.log 2>&1; echo "probe rc=$?"
grep "host api\|hub:" $OUT/host_api.log | head -40
