#!/bin/bash
# Runs on the GPU box (via gpurun): collects the evidence committed under profiles/ (round-stamped copies are made by
# the caller).
#   1. rocprofv3 --kernel-trace --stats of the bench command itself (default batch, 2 timed steps, oracle check and host-api
#      measurement skipped)
#   2. PMC passes (one counter group per pass, --kernel-trace only) of one 1920x1080x64 chunk -- the size the bench times;
#      its 796 MB intermediate does not fit the 256 MiB Infinity Cache (round 2 collected them on 16 frames) -- for HBM traffic
#   3. SQ counter passes of the same chunk
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/final
rm -rf $OUT gpurun_out/sq
mkdir -p $OUT gpurun_out/sq
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 2 --warmup 1 --no-verify --no-host-api > $OUT/bench_profiled.json 2> $OUT/stats.log
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python scripts/profile_run.py 64 > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python scripts/profile_run.py 64 > $OUT/pmc_write.log 2>&1
echo "write done"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/sq/p$i -- python scripts/profile_run.py 64 > gpurun_out/sq/p$i.log 2>&1
  echo "sq pass $i done"
done
python scripts/make_traffic_json.py 64 | cut -c1-400
python scripts/make_sq_json.py 64 | cut -c1-600
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv
