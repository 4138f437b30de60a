#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3f; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=12 > $OUT/gputests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"
tail -22 $OUT/gputests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -12 $OUT/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3f/bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'batch_bit_exact')}, d['config']['chunks_per_gpu'], d['config']['hbm_free_after_timed_steps_gb'], d['config']['sizing'])
print(d['roofline']['frac'], d['roofline_inverse']['frac'], d['entropy_chain'])
print(d.get('host_api'))
PY
