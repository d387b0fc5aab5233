#!/bin/bash
# inverse kernel: 2 segments per workgroup (default) vs 1 (XL_EXP_POLY_EXP=32); parity of the polyphase tests first
OUT=$GRAFT_REPO_ROOT/gpurun_out/s61; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "poly or random" 2>&1 | tail -3
cd /tmp
for N in 1024 4096 256; do for E in 32 0; do for SL in "8000,50000" "8000,54000"; do
XL_EXP_POLY_SLICES="$SL" XL_EXP_POLY_EXP=$E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients $N --rates 5 --modes optimized --steps 100 > $OUT/prof.log 2>&1
echo "== clients $N exp $E slices $SL: $(grep -v amdgpu $OUT/prof.log | grep optimized | awk '{print $5, $10}')"
python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
tot=0; o=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: o.append(r['Name'][4:8]+" "+str(round(float(r['AverageNs'])/1000,1))); tot+=float(r['AverageNs'])
print("   ", ", ".join(o), " sum", round(tot/1000,1))
PY
done; done; done
