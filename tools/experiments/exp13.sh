#!/bin/bash
OUT=gpurun_out/s14; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
for pr in 0 1; do echo "== NCO prio $pr"; XL_EXP_NCOPRIO=$pr python tools/sweep.py --clients 64,1024,4096 --rates 5,1 --modes optimized 2>&1 | grep -v amdgpu.ids | tee $OUT/prio$pr.log; done
