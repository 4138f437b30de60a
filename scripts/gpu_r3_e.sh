#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3e; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_bands.py -x -q -m gpu -k "capacities or foreign" > $OUT/new_tests.log 2>&1; rc=$?; echo "new tests rc=$rc"
tail -25 $OUT/new_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decoder_fuzz.py -x -q -m gpu > $OUT/parity.log 2>&1; rc=$?; echo "parity rc=$rc"
tail -8 $OUT/parity.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/sweep -- python scripts/transform_sweep.py $OUT/sweep.json 1 80 > $OUT/sweep.log 2>&1
echo "sweep rc=$?"
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/r3e/sweep/*/*_kernel_trace.csv'):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'alice' not in n: continue
        g = int(r.get('Grid_Size', 0))
        acc[(n.split('(')[0][:70], g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in sorted(acc.items()):
        if k[1] > 1000000: print(f"{k[0]:72s} grid {k[1]:>9d} n {len(v):3d} avg_us {sum(v) / len(v):8.1f} min {min(v):8.1f}")
PY
grep "band_kb" $OUT/sweep.log | tail -12
