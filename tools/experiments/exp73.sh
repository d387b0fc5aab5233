#!/bin/bash
# forward + mix as one launch (XL_EXP_POLY_FUSE=1, in-kernel wait on the shared spectra) vs two
OUT=$GRAFT_REPO_ROOT/gpurun_out/s73; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "poly or random" 2>&1 | tail -5
python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, sdr_server_amd as xl, siggen
taps = xl.create_low_pass_filter(1.0, 2016000, 24000, 9600)[1]
eng = xl.BatchEngine(2016000, "cu8", 262144)
for c in range(1024): eng.add_client(42, taps, -984000 + 1920*c)
for k in range(20): eng.process_host(siggen.xs_u8(k, 262144), "optimized")
eng.fetch(); print(eng.describe()); eng.close()
PY
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
run "two launches" XL_EXP_POLY_FUSE=0
run "fused" XL_EXP_POLY_FUSE=1
run "fused slices x,44000" XL_EXP_POLY_FUSE=1 XL_EXP_POLY_SLICES=8000,44000
run "fused slices x,54000" XL_EXP_POLY_FUSE=1 XL_EXP_POLY_SLICES=8000,54000
run "fused noskip" XL_EXP_POLY_FUSE=1 XL_EXP_POLY_EXP=16
done
