#!/bin/bash
# round 5, session g: ASan + UBSan build of the engine's host code on the GPU (tools/sanitize_gpu.sh), passes per workgroup of the
# two-half mix with 16-segment passes, irregular block lengths.  Usage: gpurun --timeout 1800 -- 'bash tools/gpu_r05g.sh r05g'
TAG=${1:-r05g}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== ASan + UBSan, host code of the engine on the GPU"
( time bash tools/sanitize_gpu.sh run $OUT ) 2>&1 | tail -25
echo "== two-half mix: passes per workgroup (16-segment passes: 14 passes per 8-block call)"
for pp in 4 7 8 14; do
  timeout 300 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 --opt mix_passes_per_workgroup=$pp 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/pp=$pp /" | tee -a $OUT/mix_mfma_pp.txt
done
echo "== irregular block lengths"
timeout 600 python tools/ragged_blocks.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ragged_blocks.txt
