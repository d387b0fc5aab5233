#!/bin/bash
# round 5, session i: ASan + UBSan over the engine's host code on the GPU (tests that stay off torch); inverse kernels 5 / 3 alternating for
# ONE block per call and for 8.  Usage: gpurun --timeout 1800 -- 'bash tools/gpu_r05i.sh r05i'
TAG=${1:-r05i}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== ASan + UBSan, host code of the engine on the GPU"
( time bash tools/sanitize_gpu.sh run $OUT ) 2>&1 | tail -12
echo "== inverse kernels alternating"
for rnd in 1 2; do for inv in 5 3; do
  timeout 300 python tools/group_sweep.py --clients 128,1024,2048,4096 --groups 1 --modes optimized --poly3 --blocks 320 --opt inverse_kernel=$inv 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/inv=$inv /" | tee -a $OUT/ab_inverse_one_block.txt
  timeout 300 python tools/group_sweep.py --clients 1024 --groups 2,4,8 --modes optimized --poly3 --blocks 320 --opt inverse_kernel=$inv 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/inv=$inv /" | tee -a $OUT/ab_inverse_one_block.txt
done; done
