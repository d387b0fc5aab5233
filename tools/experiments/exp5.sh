#!/bin/bash
OUT=gpurun_out/s5; mkdir -p $OUT
echo "== pytest"; timeout 600 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -5 | tee $OUT/pytest.log
echo "== sweep"; python tools/sweep.py --clients 64,512,1024,2048,4096 --rates 5,1 --modes optimized 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.log
