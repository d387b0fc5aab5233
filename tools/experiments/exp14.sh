#!/bin/bash
OUT=gpurun_out/s15; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
echo "== quarters"; python tools/sweep.py --clients 64,1024,4096 --rates 5,1 --modes optimized 2>&1 | grep -v amdgpu.ids | tee $OUT/q.log
echo "== 1/2 3/4 7/8"; XL_EXP_PRIO78=1 python tools/sweep.py --clients 1024,2048 --rates 5,1 --modes optimized 2>&1 | grep -v amdgpu.ids | tee $OUT/p78.log
echo "== trace 1024"; XL_EXP_TRACE=$OUT/t.bin python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 3 2>&1 | grep -v amdgpu.ids | tail -1; python tools/trace_analyze.py $OUT/t.bin | head -12; rm -f $OUT/t.bin
echo "== trace 1024 p78"; XL_EXP_PRIO78=1 XL_EXP_TRACE=$OUT/t.bin python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 3 2>&1 | grep -v amdgpu.ids | tail -1; python tools/trace_analyze.py $OUT/t.bin | head -12; rm -f $OUT/t.bin
