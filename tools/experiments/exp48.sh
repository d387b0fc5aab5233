#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s48; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for SL in "6000,42000" "29500,29600" "32768,32800" "24000,24100" "24000,32000"; do
XL_EXP_POLY_SLICES="$SL" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 100 > $OUT/prof.log 2>&1
echo "== slices $SL"; grep -v amdgpu $OUT/prof.log | grep optimized
python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: print("   ", r['Name'][:30].ljust(30), r['Calls'], r['AverageNs']); tot+=float(r['AverageNs'])
print("    sum", tot)
PY
done
