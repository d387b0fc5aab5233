#!/bin/bash
# tools/gpu_r04s.sh -- round 4, session s: the 8-lanes-per-column inverse launch (inverse_kernel=5): parity, then us per block
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r04s; mkdir -p $OUT
export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_batch_gpu.py -q -x -k "8-lane or 8lane" 2>&1 | tail -5
for v in 5 6; do XL_EXP_INV=$v timeout 300 python -m pytest tests/test_batch_gpu.py -q -x -k "group_2048_clients_all or group_4096_clients_sampled or churn or staggered" 2>&1 | tail -2; done
for ik in 3 5 6 3 5 6; do echo "== inverse_kernel=$ik"
  timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --blocks 160 --poly3 --opt inverse_kernel=$ik 2>&1 | grep "^optimized"
  timeout 100 python tools/group_sweep.py --clients 1024 --groups 1 --blocks 320 --poly3 --opt inverse_kernel=$ik 2>&1 | grep "^optimized"
done
} 2>&1 | tee $OUT/inverse8.txt | cut -c1-200
