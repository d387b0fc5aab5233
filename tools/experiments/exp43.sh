#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s43; mkdir -p $OUT
echo "== pytest polyphase"; timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=300 -k "polyphase" 2>&1 | tail -3
export TMPDIR=/tmp; cd /tmp
for SL in "1,2" "6000,42000"; do for n in 1024 4096; do
XL_EXP_POLY_SLICES="$SL" XL_EXP_POLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients $n --rates 5 --modes optimized --steps 50 > $OUT/prof.log 2>&1
echo "== $n clients, slices $SL"; grep -v amdgpu $OUT/prof.log | grep optimized
python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: print("   ", r['Name'][:30].ljust(30), r['Calls'], r['AverageNs'])
PY
done; done
