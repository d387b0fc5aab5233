#!/bin/bash
OUT=gpurun_out/s22; mkdir -p $OUT
for v in "" "XL_EXP_NOFUSE=1"; do echo "== trace 1024 $v"; env $v XL_EXP_TRACE=$OUT/t.bin python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 3 2>&1 | grep -v amdgpu.ids | tail -1; python tools/trace_analyze.py $OUT/t.bin | tail -9; rm -f $OUT/t.bin; done
