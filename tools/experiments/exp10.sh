#!/bin/bash
OUT=gpurun_out/s10; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
echo "== prio quartiles (default)"; python tools/sweep.py --clients 64,512,1024,2048,4096 --rates 5,1 --modes optimized --depths 1 2>&1 | grep -v amdgpu.ids | tee $OUT/prio.log
echo "== flat prio"; XL_EXP_FLATPRIO=1 python tools/sweep.py --clients 1024,4096 --rates 5,1 --modes optimized --depths 1 2>&1 | grep -v amdgpu.ids | tee $OUT/flat.log
for h in 8 10; do echo "== H=$h prio"; XL_EXP_H=$h python tools/sweep.py --clients 1024,2048,4096 --rates 5 --modes optimized --depths 1 2>&1 | grep -v amdgpu.ids | tail -3; done
echo "== trace 1024"; XL_EXP_TRACE=$OUT/t.bin python tools/sweep.py --clients 1024 --rates 5 --modes optimized --depths 1 --steps 3 2>&1 | grep -v amdgpu.ids | tail -1; python tools/trace_analyze.py $OUT/t.bin | head -12; rm -f $OUT/t.bin
echo "== native"; python tools/sweep.py --clients 1024 --rates 5,1 --modes native --depths 1 2>&1 | grep -v amdgpu.ids | tail -2
