#!/bin/bash
OUT=gpurun_out/s29; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2; do
for r in 1 0; do echo "== riders $r"; XL_EXP_RIDERS=$r python tools/sweep.py --clients 512,768,960,1000,1024,1536,2048,4096 --rates 5 --modes optimized --steps 200 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep_riders$r.log; done
done
