#!/bin/bash
# what hosting the NCO chain costs the three polyphase launches: XL_EXP_NOFUSE=1 tabulates the phases in a launch of its
# own (serial, slow), so the per-kernel averages of forward / mix / inverse are chain-free
OUT=$GRAFT_REPO_ROOT/gpurun_out/s74; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { # label, env...
  L=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients $N --rates 5 --modes optimized --steps 100 > $OUT/prof.log 2>&1
  echo "== clients $N $L: $(grep -v amdgpu $OUT/prof.log | grep optimized | awk '{print $5, $10}')"
  python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
tot=0; o=[]
for r in csv.DictReader(open(sys.argv[1])):
    if ('xlp' in r['Name'] or 'nco' in r['Name']) and 'tables' not in r['Name']: o.append(r['Name'].replace('void ','')[3:10]+" "+str(round(float(r['AverageNs'])/1000,1))); tot+=float(r['AverageNs'])
print("   ", ", ".join(o), " sum", round(tot/1000,1))
PY
}
for N in 1024 4096 256; do
run "hosted" XL_EXP_X=0
run "chain-free launches" XL_EXP_NOFUSE=1
run "chain-free launches, no skip" XL_EXP_NOFUSE=1 XL_EXP_POLY_EXP=16 XL_EXP_INVSKIP=0
done
