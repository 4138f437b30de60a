#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of one 1920x1080xF chunk (default 64 frames).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python scripts/profile_run.py ${1:-64} ${2:-1} > $OUT/run.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/kstats/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'alice' in r['Name']:
            print(f"{r['Name'].split('(')[0][:70]:72s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
