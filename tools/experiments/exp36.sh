#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s36; mkdir -p $OUT
timeout 120 sdr-server_amd/build/ubench_chain2 | tee $OUT/ubench_chain2.txt
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
export TMPDIR=/tmp; cd /tmp
for n in 1024; do
XL_EXP_POLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$n -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients $n --rates 5 --modes optimized --steps 50 > $OUT/prof$n.log 2>&1
echo "== $n clients"; grep -v amdgpu $OUT/prof$n.log | grep optimized
python3 - $OUT/prof$n/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'xl' in r['Name'] and 'tables' not in r['Name']: print("   ", r['Name'][:30].ljust(30), r['Calls'], r['AverageNs'])
PY
done
