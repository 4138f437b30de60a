#!/bin/bash
# Full GPU suite + one-rank rehearsal of the bench's distributed path (both gather sinks) + the loud failure of --gpus 2 on a 1-GPU box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/rehearsal; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $OUT/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
for sink in device host; do
  ALICE_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 200 python bench.py --gpus 1 --steps 2 --warmup 1 --chunks 6 --no-host-api --gather-sink $sink > $OUT/dist_$sink.json 2> $OUT/dist_$sink.err; echo "dist $sink rc=$?"
  python -c "
import json,sys
b=json.loads(open('$OUT/dist_$sink.json').read().strip().splitlines()[-1]); print(b['value'], b.get('gather'), b['cpu_baseline'].get('gpu_batch_chunk0_bit_exact'))"
done
python bench.py --gpus 2 --steps 1 --warmup 0 > $OUT/gpus2.out 2> $OUT/gpus2.err; echo "gpus2 rc=$? (non-zero expected)"; tail -2 $OUT/gpus2.err
