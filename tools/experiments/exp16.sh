#!/bin/bash
OUT=gpurun_out/s17; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
echo "== fused NCO role"; python tools/sweep.py --clients 64,512,1024,2048,4096 --rates 5,1 --modes optimized 2>&1 | grep -v amdgpu.ids | tee $OUT/fused.log
echo "== NOFUSE"; XL_EXP_NOFUSE=1 python tools/sweep.py --clients 1024 --rates 5,1 --modes optimized 2>&1 | grep -v amdgpu.ids | tail -2
echo "== native"; python tools/sweep.py --clients 1024 --rates 5,1 --modes native 2>&1 | grep -v amdgpu.ids | tail -2
echo "== bench"; timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cut -c1-700 $OUT/bench.json; tail -3 $OUT/bench.err
