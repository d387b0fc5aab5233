#!/bin/bash
# round 5, session w: the reservation rule as shipped (none from 2049 to 3008 clients) across client counts; the whole GPU suite
TAG=${1:-r05w}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/group_sweep.py --clients 1024,1536,2048,2304,2560,2816,3072,4096 --groups 8 --modes optimized --blocks 320 2>&1 | grep optimized | tee $OUT/sweep_clients.txt
timeout 900 python -m pytest tests -m gpu -q --timeout=600 2>&1 | tail -3 | tee $OUT/pytest_gpu_tail.txt
