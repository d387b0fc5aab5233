#!/bin/bash
OUT=gpurun_out/s23; mkdir -p $OUT
for n in 1024 960; do echo "== trace $n"; XL_EXP_TRACE=$OUT/t.bin python tools/sweep.py --clients $n --rates 5 --modes optimized --steps 3 2>&1 | grep -v amdgpu.ids | tail -1; python tools/trace_analyze.py $OUT/t.bin | tee $OUT/trace_$n.txt; rm -f $OUT/t.bin; done
