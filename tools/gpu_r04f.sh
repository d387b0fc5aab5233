#!/bin/bash
# round 4, session f: pipelined one-block calls (two engine compute streams), the role soak, the new multi-host timing test; then
# one block per call, pipelined vs not, 128 / 1024 / 2048 / 4096 clients.
TAG=${1:-r04f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest new tests"
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=400 -k "pipelined or soak or feed_timing or c_multi_host or feed_done" > $OUT/pytest_new.txt 2>&1
rc=$?; tail -6 $OUT/pytest_new.txt; if [ $rc -ne 0 ]; then tail -40 $OUT/pytest_new.txt; fi
echo "== one block per call: pipelined"
timeout 300 python tools/group_sweep.py --clients 128,1024,2048,4096 --groups 1 --blocks 400 --opt pipeline_calls=1 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_pipe1.txt
echo "== one block per call: not pipelined"
timeout 300 python tools/group_sweep.py --clients 128,1024,2048,4096 --groups 1 --blocks 400 --opt pipeline_calls=0 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_pipe0.txt
