#!/bin/bash
# round 5, session h: ASan + UBSan over the engine's host code on the GPU (gcc runtime), irregular block lengths with the adaptive
# look-ahead, the tests around the look-ahead.  Usage: gpurun --timeout 1800 -- 'bash tools/gpu_r05h.sh r05h'
TAG=${1:-r05h}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== ASan + UBSan, host code of the engine on the GPU"
( time bash tools/sanitize_gpu.sh run $OUT ) 2>&1 | tail -25
echo "== irregular block lengths"
timeout 600 python tools/ragged_blocks.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ragged_blocks.txt
echo "== look-ahead tests"
( time timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=600 -k "chain or nco_tabulation or side_and_fused or ragged or drift or one_block" ) 2>&1 | tail -5
