#!/bin/bash
OUT=gpurun_out/s4; mkdir -p $OUT
echo "== pytest"; timeout 600 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -5 | tee $OUT/pytest.log
echo "== KT=2 (default)"; python tools/sweep.py --clients 64,512,1024,2048,4096 --rates 5,1 --modes optimized,native 2>&1 | grep -v amdgpu.ids | tee $OUT/kt2.log
echo "== KT=1"; XL_EXP_KT=1 python tools/sweep.py --clients 1024,4096 --rates 5,1 2>&1 | grep -v amdgpu.ids | tee $OUT/kt1.log
echo "== KT=2 YFAST"; XL_EXP_YFAST=1 python tools/sweep.py --clients 1024,4096 --rates 5,1 2>&1 | grep -v amdgpu.ids | tee $OUT/kt2y.log
