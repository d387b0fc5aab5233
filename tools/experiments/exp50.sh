#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s50; mkdir -p $OUT
XL_EXP_POLY_TRACE=$OUT/t.bin python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 30 2>&1 | grep -v amdgpu.ids | tail -1; python tools/poly_trace.py $OUT/t.bin 16 2048 | tee $OUT/trace.txt
