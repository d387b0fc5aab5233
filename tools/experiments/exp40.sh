#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s40; mkdir -p $OUT
echo "== pytest polyphase"; timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=300 -k "polyphase" 2>&1 | tail -3
timeout 600 python tools/poly_check.py 64,128,256,384,512,768 5 2>&1 | grep -v amdgpu.ids | grep -v "^   " | tee $OUT/poly_clients.log
timeout 600 python tools/poly_check.py 1024,2048 1,2,3,4,5 2>&1 | grep -v amdgpu.ids | grep -v "^   " | tee $OUT/poly_taps.log
