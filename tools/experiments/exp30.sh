#!/bin/bash
OUT=gpurun_out/s30; mkdir -p $OUT
echo "== pytest polyphase"; timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=300 -k "polyphase" 2>&1 | tail -25 | tee $OUT/pytest_poly.log
echo "== poly check"; timeout 600 python tools/poly_check.py 256,1024,4096 2>&1 | grep -v amdgpu.ids | tee $OUT/poly_check.log
