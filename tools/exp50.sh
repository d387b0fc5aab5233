#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s50; mkdir -p $OUT
for n in 1024; do XL_EXP_POLY_TRACE=$OUT/t_$n.txt python tools/sweep.py --clients $n --rates 5 --modes optimized --steps 30 2>&1 | grep -v amdgpu.ids | tail -1; cat $OUT/t_$n.txt; done
