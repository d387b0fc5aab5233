#!/bin/bash
# round 5, session p: CUs reserved for the chain kernel at 4096 / 3072 clients: 8 per XCD (one round of chain workgroups), fewer (the
# chain launch runs in rounds on them), none.
TAG=${1:-r05p}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rnd in 1 2; do
  for r in 8 6 4 3 2 0; do
    XL_EXP_RESERVE=$r XL_EXP_INV=7 timeout 200 python tools/group_sweep.py --clients 3072,4096 --groups 8 --modes optimized --blocks 480 2>&1 | grep optimized | sed "s/^/reserve=$r /"
  done
done | tee $OUT/sweep_reserve.txt
