#!/bin/bash
OUT=gpurun_out/s7; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -5 | tee $OUT/pytest.log
echo "== auto NW"; python tools/sweep.py --clients 64,512,1024,2048,4096 --rates 5,1 --modes optimized --depths 1 2>&1 | grep -v amdgpu.ids | tee $OUT/auto.log
for nw in 4 5 6 7 8; do echo "== NW=$nw"; XL_EXP_NW=$nw python tools/sweep.py --clients 1024,4096 --rates 5 --modes optimized --depths 1 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/nw$nw.log; done
echo "== native auto"; python tools/sweep.py --clients 1024 --rates 5,1 --modes native --depths 1 2>&1 | grep -v amdgpu.ids | tee $OUT/native.log
