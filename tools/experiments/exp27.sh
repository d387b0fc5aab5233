#!/bin/bash
# NCO riders (spare waves of FIR workgroups) vs NCO workgroups of their own
OUT=gpurun_out/s27; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
for r in 1 0; do echo "== riders $r"; XL_EXP_RIDERS=$r python tools/sweep.py --clients 256,512,960,1000,1024,2048,4096 --rates 5 --modes optimized,native --steps 100 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep_riders$r.log; done
echo "== riders 1, 101 taps"; python tools/sweep.py --clients 1024,2048 --rates 1 --modes optimized --steps 100 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep_101.log
for n in 1024 960; do echo "== trace $n"; XL_EXP_TRACE=$OUT/t.bin python tools/sweep.py --clients $n --rates 5 --modes optimized --steps 3 2>&1 | grep -v amdgpu.ids | tail -1; python tools/trace_analyze.py $OUT/t.bin | tee $OUT/trace_$n.txt; rm -f $OUT/t.bin; done
