#!/bin/bash
# tools/gpu_r04zz.sh -- round 4, last session: the PERSISTENT form of the 8-lane inverse launch (option inverse_persistent = workgroups per CU):
# parity of the default and of the persistent form, then us per block, alternating
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r04zz; mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== parity, default"; timeout 200 python -m pytest tests/test_batch_gpu.py -q -x -k "8-lane or group_2048_clients_all or churn" 2>&1 | tail -1
echo "== parity, inverse_persistent=4"; XL_EXP_INV_PERSIST=4 timeout 200 python -m pytest tests/test_batch_gpu.py -q -x -k "8-lane or group_2048_clients_all or group_4096_clients_sampled or churn or staggered" 2>&1 | tail -1
for p in 0 4 0 4 3 5; do echo "== inverse_persistent=$p"
  timeout 100 python tools/group_sweep.py --clients 2048,4096 --groups 8 --blocks 160 --opt inverse_persistent=$p 2>&1 | grep "^optimized"
done
} 2>&1 | tee $OUT/inverse8_persistent.txt | cut -c1-160
