#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s51; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for PR in 3 2 1 0; do
XL_EXP_NCOPRIO=$PR timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 100 > $OUT/prof.log 2>&1
echo "== NCO prio $PR"; grep -v amdgpu $OUT/prof.log | grep optimized
python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: print("   ", r['Name'][:30].ljust(30), r['Calls'], r['AverageNs']); tot+=float(r['AverageNs'])
print("    sum", tot)
PY
done
cd $GRAFT_REPO_ROOT
for PR in 3 0; do echo "== direct, NCO prio $PR"; XL_EXP_NCOPRIO=$PR XL_EXP_POLY=0 python tools/sweep.py --clients 1024 --rates 5,1 --modes optimized,native --steps 100 2>&1 | grep -v amdgpu.ids | grep -v "^mode"; done
