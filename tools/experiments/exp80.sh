#!/bin/bash
# mix passes of 7 segments (XL_EXP_POLY_SEG=7: twice the waves, half the accumulators) vs 14
OUT=$GRAFT_REPO_ROOT/gpurun_out/s80; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "poly or random" 2>&1 | tail -3
cd /tmp
run() { # label, env...
  L=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients $N --rates 5 --modes optimized --steps 200 > $OUT/prof.log 2>&1
  python3 - $OUT/prof/p_kernel_stats.csv "$N $L" "$(grep -v amdgpu $OUT/prof.log | grep optimized | awk '{print $5, $10}')" <<'PY'
import csv, sys
tot=0; o=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: o.append(r['Name'].replace('void ','')[4:8]+" "+str(round(float(r['AverageNs'])/1000,1))); tot+=float(r['AverageNs'])
print(sys.argv[2], "|", sys.argv[3], "|", ", ".join(o), " sum", round(tot/1000,1))
PY
}
for N in 1024 512 256 2048 4096; do
run "seg14" XL_EXP_POLY_SEG=14
run "seg7" XL_EXP_POLY_SEG=7
run "seg7 mixskip 2048" XL_EXP_POLY_SEG=7 XL_EXP_MIXSKIP=2048
run "seg7 noskip" XL_EXP_POLY_SEG=7 XL_EXP_POLY_EXP=16
done
