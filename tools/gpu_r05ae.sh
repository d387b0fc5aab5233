#!/bin/bash
# round 5, session ae: the reservation rule as shipped (by load; rounds only for plans of about the measured shape's weight): config 5 and the default shape, and the tests that pin the plan's choices
TAG=${1:-r05ae}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rnd in 1 2; do
  timeout 200 python tools/group_sweep.py --shape config5 --clients 512,1024,2048,4096 --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized
done | tee $OUT/sweep_config5.txt
timeout 200 python tools/group_sweep.py --clients 2048,2304,4096 --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | tee $OUT/sweep_default.txt
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -k "2304 or expected_clients or 4096_clients_sampled or config5" --timeout=300 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
