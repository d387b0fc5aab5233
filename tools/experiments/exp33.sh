#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s33; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for e in 0 1 2 3; do
for sl in "6000,42000" "1,2"; do
XL_EXP_POLY_SLICES=$sl XL_EXP_POLY_EXP=$e XL_EXP_POLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$e -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients 4096 --rates 5 --modes optimized --steps 30 > $OUT/prof$e.log 2>&1
echo "== exp $e slices $sl"; grep -v amdgpu $OUT/prof$e.log | grep optimized
python3 - $OUT/prof$e/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: print("   ", r['Name'][:30].ljust(30), r['Calls'], r['AverageNs'])
PY
done
done
