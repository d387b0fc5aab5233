#!/bin/bash
# round 5, session k: the 32 x 4 inverse kernel (xl_inv32.hip, inverse_kernel 6 / 7 / 8 = 1 / 2 / 4 waves per workgroup): parity on the
# forced-path tests, then against the LDS transform (3) and the 8-lane kernel (5), alternating in one process per shape.
TAG=${1:-r05k}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity (forced polyphase tests, cut32 fixtures)"
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "cut32" --timeout=300 2>&1 | tail -5 | tee $OUT/pytest_cut32.txt
echo "== sweeps"
for rnd in 1 2; do
  for inv in 3 5 6 7 8; do
    XL_EXP_INV=$inv timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/inv=$inv /"
  done
done | tee $OUT/sweep_inv.txt
for inv in 3 5 6 7; do
  XL_EXP_INV=$inv timeout 200 python tools/group_sweep.py --clients 1024,2048 --groups 1 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/inv=$inv /"
done | tee $OUT/sweep_inv_g1.txt
