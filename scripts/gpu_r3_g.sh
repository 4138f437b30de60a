#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3g; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 python scripts/host_api_probe.py $OUT/host_api.json 1 8 64 > $OUT/host_api.log 2>&1; echo "host api rc=$?"
tail -5 $OUT/host_api.log | cut -c1-600
# two chains per SIMD: with and without the priority alternation (960x540x64 chunks so that 1200 and 2046 chains fit)
PROBE_PLACEMENT=1 timeout -k 10 500 python scripts/chain_probe.py 960 540 64 341 400 682 > $OUT/chain_turns_on.log 2>&1; echo "chains on rc=$?"
grep -E "chunks|shared" $OUT/chain_turns_on.log | cut -c1-260
ALICE_CHAIN_TURNS=0 PROBE_PLACEMENT=1 timeout -k 10 500 python scripts/chain_probe.py 960 540 64 400 682 > $OUT/chain_turns_off.log 2>&1; echo "chains off rc=$?"
grep -E "chunks|shared" $OUT/chain_turns_off.log | cut -c1-260
