#!/bin/bash
# Runs on the GPU box (via gpurun): collects the evidence committed under profiles/.
#   1. rocprofv3 --kernel-trace --stats of the bench command itself (default batch, 2 timed steps, oracle check skipped)
#   2. PMC passes (one counter group per pass, --kernel-trace only) of one 1920x1080x16 chunk for HBM traffic
#   3. the headline bench line at the default batch
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/final
rm -rf $OUT
mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 2 --warmup 1 --no-verify > $OUT/bench_profiled.json 2> $OUT/stats.log
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python scripts/profile_run.py 16 > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python scripts/profile_run.py 16 > $OUT/pmc_write.log 2>&1
echo "write done"
timeout -k 10 600 python bench.py --steps 2 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench done"
tail -c 400 $OUT/bench_default.json
