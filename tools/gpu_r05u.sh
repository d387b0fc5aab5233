#!/bin/bash
# round 5, session u: the final tree across client counts (8 blocks per call), and the chain reservation rule where it extrapolates
# (5120 / 6144 / 8192 clients: 3 / 3 / 4 rounds) against fewer rounds (XL_EXP_RESERVE cannot ask for MORE CUs than the rule; XL_EXP_ROUNDS1
# restores one CU per chain workgroup) and none.
TAG=${1:-r05u}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/group_sweep.py --clients 128,256,512,1024,1536,2048,2560,3072,4096,5120,6144,8192 --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | tee $OUT/sweep_clients.txt
for rnd in 1 2; do
  for c in 5120 6144 8192; do
    timeout 200 python tools/group_sweep.py --clients $c --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/rule      /"
    XL_EXP_ROUNDS1=1 timeout 200 python tools/group_sweep.py --clients $c --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/one round /"
    XL_EXP_NOMASK=1 timeout 200 python tools/group_sweep.py --clients $c --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/no mask   /"
  done
done | tee $OUT/sweep_rounds.txt
timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 1,2,4,8,16 --modes optimized --blocks 320 2>&1 | grep optimized | tee $OUT/sweep_blocks_per_call.txt
