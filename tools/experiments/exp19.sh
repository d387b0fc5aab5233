#!/bin/bash
OUT=gpurun_out/s21; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -2
echo "== sweep"; python tools/sweep.py --clients 64,512,960,1024,2048,4096 --rates 5,1 --modes optimized 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.log
for n in 1024; do echo "== trace $n"; XL_EXP_TRACE=$OUT/t.bin python tools/sweep.py --clients $n --rates 5 --modes optimized --steps 3 2>&1 | grep -v amdgpu.ids | tail -1; python tools/trace_analyze.py $OUT/t.bin | tee $OUT/trace_$n.txt | head -12; rm -f $OUT/t.bin; done
