#!/bin/bash
# round 5, session m: the 32 x 4 inverse kernel, row-split form (120 registers: four waves per SIMD; 128-byte store runs) against the
# column-split form (154 registers: three; 256-byte runs) and the LDS transform, alternating.
TAG=${1:-r05m}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity (forced polyphase tests, cut32 fixtures)"
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "cut32" --timeout=300 2>&1 | tail -3 | tee $OUT/pytest_cut32.txt
echo "== sweeps"
for rnd in 1 2; do
  for inv in 3 7 9 10 11; do
    XL_EXP_INV=$inv timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/inv=$inv /"
  done
done | tee $OUT/sweep_inv.txt
for inv in 3 5 7 10; do
  XL_EXP_INV=$inv timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 1 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/inv=$inv /"
done | tee $OUT/sweep_inv_g1.txt
