#!/bin/bash
# A/B of transform builds under rocprofv3 (ab_libs/*.so named on the command line) + parity of the in-tree build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ab; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bands.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -5 $OUT/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
for L in "$@"; do
  export ALICE_CODEC_LIB=$PWD/ab_libs/$L
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/$L -- python scripts/ab_transform.py 20 > $OUT/$L.log 2>&1 || { echo "$L failed"; tail -5 $OUT/$L.log; exit 1; }
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/ab/*/*/*_kernel_trace.csv')):
    print(f.split('/')[2])
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'alice' not in n: continue
        acc[(n.split('(')[0][:70], r['VGPR_Count'], r['LDS_Block_Size'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in sorted(acc.items()):
        print(f"  {k[0]:72s} vgpr {k[1]:>3s} lds {k[2]:>6s} n {len(v):3d} avg_us {sum(v) / len(v):8.1f} min {min(v):8.1f}")
PY
