#!/bin/bash
# EXPERIMENT (timing only, racy tables): the NCO chain as a kernel on a side stream, concurrent with the chain-free
# polyphase launches.  XL_EXP_SIDECHAIN=1: 16 one-wave workgroups; 2: 4 workgroups of 4 waves holding all of a CU's LDS
# (no LDS-using workgroup shares the CU); 3: the same without the LDS allocation
OUT=$GRAFT_REPO_ROOT/gpurun_out/s75; mkdir -p $OUT
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
    if ('xlp' in r['Name'] or 'nco' in r['Name']) and 'tables' not in r['Name']: o.append(r['Name'].replace('void ','')[3:10]+" "+str(round(float(r['AverageNs'])/1000,1))+" x"+r['Calls']); tot+=float(r['AverageNs'])
print("   ", ", ".join(o))
PY
}
for N in 1024 4096 256; do
run "hosted" XL_EXP_X=0
run "side 1" XL_EXP_NOFUSE=1 XL_EXP_SIDECHAIN=1 XL_EXP_POLY_EXP=16 XL_EXP_INVSKIP=0
run "side 2 (CU monopoly)" XL_EXP_NOFUSE=1 XL_EXP_SIDECHAIN=2 XL_EXP_POLY_EXP=16 XL_EXP_INVSKIP=0
run "side 3 (4-wave wgs)" XL_EXP_NOFUSE=1 XL_EXP_SIDECHAIN=3 XL_EXP_POLY_EXP=16 XL_EXP_INVSKIP=0
done
