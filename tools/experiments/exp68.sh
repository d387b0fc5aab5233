#!/bin/bash
# mix: one-wave workgroups again, the passes of one (m, cg) 8 grid positions apart (same XCD); M = 128 vs 256; PMC traffic
OUT=$GRAFT_REPO_ROOT/gpurun_out/s68; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "poly or random" 2>&1 | tail -3
cd /tmp
run() { # label, env...
  L=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients $N --rates 5 --modes optimized --steps 100 > $OUT/prof.log 2>&1
  echo "== clients $N $L: $(grep -v amdgpu $OUT/prof.log | grep optimized | awk '{print $5, $10}')"
  python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
tot=0; o=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: o.append(r['Name'][4:8]+" "+str(round(float(r['AverageNs'])/1000,1))); tot+=float(r['AverageNs'])
print("   ", ", ".join(o), " sum", round(tot/1000,1))
PY
}
for N in 1024 4096 2048; do
run "M256" XL_EXP_POLY_M=256
run "M128" XL_EXP_POLY_M=128
run "M128 slices 8000,44000" XL_EXP_POLY_M=128 XL_EXP_POLY_SLICES=8000,44000
run "M128 noskip" XL_EXP_POLY_M=128 XL_EXP_POLY_EXP=16
done
cd $GRAFT_REPO_ROOT
XL_EXP_POLY_M=128 XL_EXP_POLY_TRACE=$OUT/t.bin python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 3 2>&1 | grep optimized
python tools/poly_place.py $OUT/t.bin 2048; python tools/poly_trace.py $OUT/t.bin 16 2048 | grep -v "^nco wave"
rm -f $OUT/t.bin
bash tools/pmc_traffic.sh s68 > $OUT/pmc.log 2>&1; python3 -c "
import json; d=json.load(open('$OUT/pmc_traffic.json')); print(json.dumps(d['poly'],indent=0))"
