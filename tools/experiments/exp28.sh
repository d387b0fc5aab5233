#!/bin/bash
OUT=gpurun_out/s28; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
for r in 1; do echo "== riders $r"; XL_EXP_RIDERS=$r python tools/sweep.py --clients 128,256,384,512,768,960,1000,1024,1536,2048,3000,4096 --rates 5 --modes optimized,native --steps 100 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep_riders$r.log; done
for r in 1; do echo "== riders $r, 101 taps"; XL_EXP_RIDERS=$r python tools/sweep.py --clients 512,1024,2048,4096 --rates 1 --modes optimized --steps 100 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep_101_$r.log; done
echo "== riders forced at small sizes"; XL_EXP_RIDERS_MIN=1 python tools/sweep.py --clients 128,256,384 --rates 5 --modes optimized --steps 100 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep_small.log
