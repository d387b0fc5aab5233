#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s60; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for SL in "6000,42000" "6000,47000" "6000,52000" "3000,50000" "8000,50000" "6000,56000"; do
XL_EXP_POLY_SLICES="$SL" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 100 > $OUT/prof.log 2>&1
echo "== slices $SL: $(grep -v amdgpu $OUT/prof.log | grep optimized | awk '{print $5, $10}')"
python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
tot=0; o=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: o.append(r['Name'][4:8]+" "+str(round(float(r['AverageNs'])/1000,1))); tot+=float(r['AverageNs'])
print("   ", ", ".join(o), " sum", round(tot/1000,1))
PY
done
