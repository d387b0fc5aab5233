#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s52; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for E in 0 4 8; do
XL_EXP_POLY_EXP=$E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 100 > $OUT/prof.log 2>&1
echo "== exp $E"; grep -v amdgpu $OUT/prof.log | grep optimized
python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: print("   ", r['Name'][:30].ljust(30), r['Calls'], r['AverageNs'])
PY
done
