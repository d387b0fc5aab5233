#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s56; mkdir -p $OUT
echo "== pytest polyphase"; timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=300 -k "polyphase" 2>&1 | tail -2
XL_EXP_POLY_TRACE=$OUT/t.bin python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 30 2>&1 | grep -v amdgpu.ids | tail -1; python tools/poly_trace.py $OUT/t.bin 16 2048 | grep -v "^nco wave 1[0-9]\|^nco wave [2-9]" | tee $OUT/trace.txt
export TMPDIR=/tmp; cd /tmp
for E in 0 32; do
XL_EXP_POLY_EXP=$E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 100 > $OUT/prof.log 2>&1
echo "== exp $E: $(grep -v amdgpu $OUT/prof.log | grep optimized | awk '{print $5, $10}')"
python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
tot=0; o=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: o.append(r['Name'][4:8]+" "+str(round(float(r['AverageNs'])/1000,1))); tot+=float(r['AverageNs'])
print("   ", ", ".join(o), " sum", round(tot/1000,1))
PY
done
