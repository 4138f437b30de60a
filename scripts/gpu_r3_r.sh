#!/bin/bash
# full suite, short bench (chain ns/symbol after the kernel's new descriptor parameter), soak
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3r; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $OUT/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "
import json
b=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(b['value'], b['entropy_chain']['encode_ns_per_symbol'], b['entropy_chain']['decode_ns_per_symbol'], b['roofline']['frac'], b['roofline_inverse']['frac']); print([ (r['threads'], r['encode_mpix_s'], r['decode_mpix_s']) for r in b['host_api']['rows']])"
timeout -k 10 300 python tests/tools/soak_gpu.py 200 30001 > $OUT/soak.log 2>&1; echo "soak rc=$?"; tail -3 $OUT/soak.log | cut -c1-400
