#!/bin/bash
# re-tune of the inverse launch's empty slot and the slice split for the M = 128 plan (1024 clients)
OUT=$GRAFT_REPO_ROOT/gpurun_out/s77; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { # label, env...
  L=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients $N --rates 5 --modes optimized --steps 200 > $OUT/prof.log 2>&1
  python3 - $OUT/prof/p_kernel_stats.csv "$N $L: $(grep -v amdgpu $OUT/prof.log | grep optimized | awk '{print $5, $10}')" <<'PY'
import csv, sys
tot=0; o=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: o.append(r['Name'].replace('void ','')[4:8]+" "+str(round(float(r['AverageNs'])/1000,1))); tot+=float(r['AverageNs'])
print(sys.argv[2], "|", ", ".join(o), " sum", round(tot/1000,1))
PY
}
N=1024
run "default (invskip 256, slices 8000,50000)" XL_EXP_X=0
run "invskip 0" XL_EXP_INVSKIP=0
run "invskip 128" XL_EXP_INVSKIP=128
run "invskip 512" XL_EXP_INVSKIP=512
run "invskip 64" XL_EXP_INVSKIP=64
run "slices 6000,50000" XL_EXP_POLY_SLICES=6000,50000
run "slices 10000,50000" XL_EXP_POLY_SLICES=10000,50000
run "slices 8000,52000" XL_EXP_POLY_SLICES=8000,52000
run "slices 8000,47000" XL_EXP_POLY_SLICES=8000,47000
run "default again" XL_EXP_X=0
