#!/bin/bash
# round 5, session ab: the size rule's upper boundary (8192 tiles): 1152 / 1280 / 1536 / 1792 clients x 8 blocks (7776 / 8640 / 10368 / 12096 tiles), LDS transform (3) against the 32 x 4 cut (6)
TAG=${1:-r05ab}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rnd in 1 2; do
  for inv in 3 6; do
    XL_EXP_INV=$inv timeout 200 python tools/group_sweep.py --clients 1152,1280,1536,1792 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/inv=$inv /"
  done
done | tee $OUT/sweep_boundary.txt
