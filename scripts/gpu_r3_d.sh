#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3d; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/sweepq -- python scripts/transform_sweep.py $OUT/sweepq.json 1 80 quick > $OUT/sweepq.log 2>&1
echo "sweepq rc=$?"
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/r3d/sweepq/*/*_kernel_trace.csv'):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'alice' not in n: continue
        g = int(r.get('Grid_Size', 0))
        if g < 1500000: continue
        acc[(n.split('(')[0][:70], g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in sorted(acc.items()):
        print(f"{k[0]:72s} grid {k[1]:>9d} n {len(v):3d} avg_us {sum(v) / len(v):8.1f} min {min(v):8.1f}")
PY
grep "probe" $OUT/sweepq.log | head -12
