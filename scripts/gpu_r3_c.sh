#!/bin/bash
# round 3, GPU call C: A/B of the round-2 kernels, the round-2 kernels with the LDS layout fix, and the new build
# (uncut, one role per launch); per-role VALU floors
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3c; rm -rf $OUT; mkdir -p $OUT
summ() { python - "$1" "$2" <<'PY'
import csv, glob, sys, collections
d, label = sys.argv[1], sys.argv[2]
for f in glob.glob(d + '/*/*_kernel_trace.csv'):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'alice' not in n: continue
        key = (n.split('(')[0][:64], r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('VGPR_Count', '?'), r.get('LDS_Block_Size', '?'))
        acc[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in sorted(acc.items()):
        v2 = v[len(v) // 4:]    # drop the warm-up quarter
        print(f"{label:12s} {k[0]:66s} grid {k[1]:>9s} vgpr {k[2]:>4s} lds {k[3]:>6s} n {len(v):3d} avg_us {sum(v2) / len(v2):8.1f} min {min(v):8.1f}")
PY
}
for lib in r2 r2_ldsfix; do
  ALICE_CODEC_LIB=$PWD/ab_libs/libalice_$lib.so timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/$lib -- python scripts/ab_transform.py 24 > $OUT/$lib.log 2>&1
  echo "$lib rc=$?"; summ $OUT/$lib $lib | tee -a $OUT/summary.txt
done
ALICE_AB_TUNING=0,0,1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/new -- python scripts/ab_transform.py 24 > $OUT/new.log 2>&1
echo "new rc=$?"; summ $OUT/new new | tee -a $OUT/summary.txt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/sweepq -- python scripts/transform_sweep.py $OUT/sweepq.json 1 80 quick > $OUT/sweepq.log 2>&1
echo "sweepq rc=$?"; summ $OUT/sweepq sweep | tee -a $OUT/summary.txt
tail -8 $OUT/sweepq.log
