#!/bin/bash
# round 5, session v: between 2048 and 3072 clients: the rule's one round of chain workgroups (one CU each) against fewer CUs / none
TAG=${1:-r05v}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rnd in 1 2; do
  for c in 2048 2304 2560 2816; do
    timeout 200 python tools/group_sweep.py --clients $c --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/rule      /"
    XL_EXP_RESERVE=3 timeout 200 python tools/group_sweep.py --clients $c --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/reserve=3 /"
    XL_EXP_RESERVE=2 timeout 200 python tools/group_sweep.py --clients $c --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/reserve=2 /"
    XL_EXP_NOMASK=1 timeout 200 python tools/group_sweep.py --clients $c --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/no mask   /"
  done
done | tee $OUT/sweep_midband.txt
