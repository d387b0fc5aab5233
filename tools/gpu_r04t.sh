#!/bin/bash
# tools/gpu_r04t.sh -- round 4, session t: the inverse launch's memory traffic ALONE (the 8-lane kernel with transform and phases compiled out:
# the same tile loads, the same output stores) against the real kernels, us per block
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r04t; mkdir -p $OUT
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
{
for cfg in "shipped 5" "inv8_copy 5" "shipped 5" "inv8_copy 5"; do set -- $cfg
  if [ $1 = shipped ]; then unset XL_TESTING XL_LIBRARY_PATH; else export XL_TESTING=1 XL_LIBRARY_PATH=$V/lib$1.so; fi
  echo "== $1, inverse_kernel=$2"
  timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --blocks 160 --poly3 --opt inverse_kernel=$2 2>&1 | grep "^optimized"
  timeout 100 python tools/group_sweep.py --clients 1024 --groups 1 --blocks 320 --poly3 --opt inverse_kernel=$2 2>&1 | grep "^optimized"
done
} 2>&1 | tee $OUT/inverse_traffic_only.txt | cut -c1-200
