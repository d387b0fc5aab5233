#!/bin/bash
# round 5, session ad: the reservation rule by the plan's load: config 5 (rule / no mask / one CU per chain workgroup) and the default shape where the bands were measured
TAG=${1:-r05ad}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rnd in 1 2; do
  timeout 200 python tools/group_sweep.py --shape config5 --clients 512,1024,2048,4096 --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/rule      /"
  XL_EXP_NOMASK=1 timeout 200 python tools/group_sweep.py --shape config5 --clients 512,1024,2048,4096 --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/no mask   /"
  XL_EXP_ROUNDS1=1 timeout 200 python tools/group_sweep.py --shape config5 --clients 512,1024,2048,4096 --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/one round /"
done | tee $OUT/sweep_config5.txt
timeout 200 python tools/group_sweep.py --clients 1024,2048,2304,3072,4096 --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | tee $OUT/sweep_default.txt
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "2304 or expected_clients or 4096_clients_sampled or config5" --timeout=300 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
