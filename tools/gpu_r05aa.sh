#!/bin/bash
# round 5, session aa: the LDS-transform inverse kernel with a register budget for 5 waves per SIMD (96 registers, 8 spilled; 5 x 32 KB of LDS
# = a CU's whole LDS) against the shipped one (107 registers, 4 waves per SIMD), where the size rule uses it
TAG=${1:-r05aa}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
for rnd in 1 2 3; do
  timeout 200 python tools/group_sweep.py --clients 1024 --groups 4,8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/shipped /"
  XL_TESTING=1 XL_LIBRARY_PATH=$V/libinv_wpe5.so timeout 200 python tools/group_sweep.py --clients 1024 --groups 4,8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/wpe5    /"
  timeout 200 python tools/group_sweep.py --shape config5 --clients 1024 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/shipped config5 /"
  XL_TESTING=1 XL_LIBRARY_PATH=$V/libinv_wpe5.so timeout 200 python tools/group_sweep.py --shape config5 --clients 1024 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/wpe5    config5 /"
done | tee $OUT/sweep_inv_wpe5.txt
