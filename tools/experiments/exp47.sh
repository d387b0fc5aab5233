#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s47; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/scratch/sf.py 2>&1 | grep -v "amdgpu\|rocprofv3"
python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("   ", r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
