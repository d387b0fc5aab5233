#!/bin/bash
OUT=gpurun_out/s8; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -5 | tee $OUT/pytest.log
echo "== auto H"; python tools/sweep.py --clients 64,512,1024,2048,4096 --rates 5,1 --modes optimized --depths 1 2>&1 | grep -v amdgpu.ids | tee $OUT/auto.log
for h in 8 9 10 12; do echo "== H=$h"; XL_EXP_H=$h python tools/sweep.py --clients 1024,2048,4096 --rates 5 --modes optimized --depths 1 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/h$h.log; done
echo "== native auto"; python tools/sweep.py --clients 1024 --rates 5,1 --modes native --depths 1 2>&1 | grep -v amdgpu.ids | tee $OUT/native.log
