#!/bin/bash
# matrix-core mix: 32 columns per wave (XLP_MT=1, 4 waves per SIMD at 1024 clients) vs 64 (XLP_MT=2); skip / no skip; slices
OUT=$GRAFT_REPO_ROOT/gpurun_out/s72; mkdir -p $OUT
export TMPDIR=/tmp
run() { # label, env...
  L=$1; shift
  cd /tmp
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients $N --rates 5 --modes optimized --steps 100 > $OUT/prof.log 2>&1
  echo "== clients $N $L: $(grep -v amdgpu $OUT/prof.log | grep optimized | awk '{print $5, $10}')"
  python3 - $OUT/prof/p_kernel_stats.csv <<'PY'
import csv, sys
tot=0; o=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'xlp' in r['Name'] and 'tables' not in r['Name']: o.append(r['Name'].replace('void ','')[4:9]+" "+str(round(float(r['AverageNs'])/1000,1))); tot+=float(r['AverageNs'])
print("   ", ", ".join(o), " sum", round(tot/1000,1))
PY
  cd $GRAFT_REPO_ROOT
}
for MT in 1 2; do
touch sdr-server_amd/csrc/xl_polyphase.hip; make -C sdr-server_amd/csrc EXTRA=-DXLP_MT=${MT}u 2>&1 | grep -E "error"
for N in 1024 4096; do
run "MT$MT skip" XL_EXP_POLY_MFMA=1
run "MT$MT noskip" XL_EXP_POLY_MFMA=1 XL_EXP_POLY_EXP=16
run "MT$MT skip slices 12000,40000" XL_EXP_POLY_MFMA=1 XL_EXP_POLY_SLICES=12000,40000
run "MT$MT noskip slices 12000,40000" XL_EXP_POLY_MFMA=1 XL_EXP_POLY_SLICES=12000,40000 XL_EXP_POLY_EXP=16
run "MT$MT skip slices 14000,36000" XL_EXP_POLY_MFMA=1 XL_EXP_POLY_SLICES=14000,36000
done; done
