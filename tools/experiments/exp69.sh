#!/bin/bash
# mix on the matrix cores (xlp_mixm_kernel): parity, then timing vs the vector-ALU kernel (XL_EXP_POLY_MFMA=0)
OUT=$GRAFT_REPO_ROOT/gpurun_out/s69; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "poly or random" 2>&1 | tail -8
cd /tmp
run() { # label, env...
  L=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients $N --rates 5 --modes optimized --steps 100 > $OUT/prof.log 2>&1
  echo "== clients $N $L: $(grep -v amdgpu $OUT/prof.log | grep optimized | awk '{print $5, $10}')"
  python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
tot=0; o=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: o.append(r['Name'].replace('void ','')[4:9]+" "+str(round(float(r['AverageNs'])/1000,1))); tot+=float(r['AverageNs'])
print("   ", ", ".join(o), " sum", round(tot/1000,1))
PY
}
for N in 1024 4096 256; do
run "M128 valu" XL_EXP_POLY_MFMA=0
run "M128 mfma" XL_EXP_POLY_MFMA=1
run "M128 mfma noskip" XL_EXP_POLY_MFMA=1 XL_EXP_POLY_EXP=16
run "M128 mfma slices 12000,40000" XL_EXP_POLY_MFMA=1 XL_EXP_POLY_SLICES=12000,40000
run "M128 mfma slices 12000,36000" XL_EXP_POLY_MFMA=1 XL_EXP_POLY_SLICES=12000,36000
run "M256 mfma" XL_EXP_POLY_MFMA=1 XL_EXP_POLY_M=256
done
