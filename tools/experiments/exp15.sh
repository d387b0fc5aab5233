#!/bin/bash
OUT=gpurun_out/s16; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
for ln in 8 16 32 64; do for pr in 3 1; do echo "== NCO lanes $ln prio $pr"; XL_EXP_NCOLANES=$ln XL_EXP_NCOPRIO=$pr python tools/sweep.py --clients 64,1024 --rates 5,1 --modes optimized 2>&1 | grep -v amdgpu.ids | tail -4; done; done
