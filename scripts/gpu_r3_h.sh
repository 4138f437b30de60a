#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3h; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 python scripts/transform_sweep.py $OUT/sweep_px4.json 1 80 quick > $OUT/sweep_px4.log 2>&1; echo "px4 rc=$?"; grep band_kb $OUT/sweep_px4.log
ALICE_CODEC_LIB=$PWD/ab_libs/libalice_px8.so timeout -k 10 200 python scripts/transform_sweep.py $OUT/sweep_px8.json 1 80 quick > $OUT/sweep_px8.log 2>&1; echo "px8 rc=$?"; grep band_kb $OUT/sweep_px8.log
ALICE_CODEC_LIB=$PWD/ab_libs/libalice_px8.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size_1080p64_cdf97 or seeded or unaligned or batch" > $OUT/px8_parity.log 2>&1; echo "px8 parity rc=$?"; tail -3 $OUT/px8_parity.log
GPU_MAX_HW_QUEUES=32 timeout -k 10 400 python scripts/host_api_probe.py $OUT/host_api_q32.json 8 64 > $OUT/host_api_q32.log 2>&1; echo "host api q32 rc=$?"
grep "host api" $OUT/host_api_q32.log
