#!/bin/bash
# round 3, GPU call A: parity of the LDS-layout change, per-kernel times, stream-overlap probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3a; rm -rf $OUT; mkdir -p $OUT
set -o pipefail
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/parity.log 2>&1; echo "parity rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/parity.log
timeout -k 10 120 python scripts/transform_probe.py 1 20 > $OUT/transform_probe.log 2>&1; echo "probe rc=$?" | tee -a $OUT/summary.txt
cat $OUT/transform_probe.log
timeout -k 10 200 python scripts/overlap_probe.py 20 $OUT/overlap.json > $OUT/overlap.log 2>&1; echo "overlap rc=$?" | tee -a $OUT/summary.txt
cat $OUT/overlap.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -- python scripts/profile_run.py 64 1 > $OUT/kstats.log 2>&1; echo "kstats rc=$?" | tee -a $OUT/summary.txt
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r3a/kstats/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'alice' in r['Name']:
            print(f"{r['Name'].split('(')[0][:70]:72s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
