#!/bin/bash
# round 5, session o: do the polyphase launches at 2048 / 4096 clients run faster on the whole chip than on the CUs the chain kernel's
# reservation leaves them (192 of 256 at 4096 clients)?  XL_EXP_NOMASK=1: no reservation (the chain kernel then shares the chip).
TAG=${1:-r05o}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rnd in 1 2; do
  for inv in 3 7; do
    XL_EXP_INV=$inv timeout 200 python tools/group_sweep.py --clients 2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/inv=$inv reserved /"
    XL_EXP_NOMASK=1 XL_EXP_INV=$inv timeout 200 python tools/group_sweep.py --clients 2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/inv=$inv nomask   /"
  done
done | tee $OUT/sweep_nomask.txt
