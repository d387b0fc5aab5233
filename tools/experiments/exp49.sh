#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s49; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -8 | tee $OUT/pytest.log
echo "== sweep"; python tools/sweep.py --clients 256,1024,2048,4096 --rates 5,1 --modes optimized,native --steps 100 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.log
echo "== direct optimized"; XL_EXP_POLY=0 python tools/sweep.py --clients 1024,4096 --rates 5 --modes optimized --steps 100 2>&1 | grep -v amdgpu.ids | grep -v "^mode" | tee $OUT/sweep_direct.log
export TMPDIR=/tmp; cd /tmp
for n in 1024 4096; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients $n --rates 5 --modes optimized --steps 100 > $OUT/prof.log 2>&1
echo "== $n clients"; grep -v amdgpu $OUT/prof.log | grep optimized
python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'xl' in r['Name'] and 'tables' not in r['Name']: print("   ", r['Name'][:30].ljust(30), r['Calls'], r['AverageNs'])
PY
done
