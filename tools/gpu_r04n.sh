#!/bin/bash
# round 4, session n: completion events thinned out (one per chain launch instead of one per call; no per-feed event in the in-place
# multi host): parity of the side-stream tests, then one block per call and 8 blocks per call.
TAG=${1:-r04n}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=400 -k "feed_done or feed_timing or c_multi_host or pipelined or drift or churn or side or group_of_blocks or 4096_clients or staggered or ncalls" > $OUT/pytest_sel.txt 2>&1
rc=$?; tail -3 $OUT/pytest_sel.txt; if [ $rc -ne 0 ]; then tail -40 $OUT/pytest_sel.txt; exit 1; fi
timeout 300 python tools/group_sweep.py --clients 128,1024,2048 --groups 1 --blocks 400 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_g1.txt
timeout 300 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --blocks 400 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_g8.txt
