#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/probe_twins; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 700 python -m pytest tests/test_gpu_value_table.py tests/test_gpu_parity.py tests/test_gpu_bands.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -5 $OUT/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/sweep -- python scripts/transform_sweep.py $OUT/sweep.json 1 80 quick > $OUT/sweep.log 2>&1; echo "sweep rc=$?"
grep band_kb $OUT/sweep.log
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/probe_twins/sweep/*/*_kernel_trace.csv'):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'alice' not in n: continue
        acc[(n.split('(')[0][:70], r['VGPR_Count'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in sorted(acc.items()):
        print(f"{k[0]:72s} vgpr {k[1]:>3s} n {len(v):3d} avg_us {sum(v) / len(v):8.1f} min {min(v):8.1f}")
PY
