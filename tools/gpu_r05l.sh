#!/bin/bash
# round 5, session l: the 32 x 4 inverse kernel in two rounds of 8 columns (9984 bytes of LDS per wave): 3 waves per SIMD (154
# registers) and 4 (128 registers, 23 spilled), against the LDS transform, alternating.
TAG=${1:-r05l}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
echo "== parity (forced polyphase tests, cut32 fixtures)"
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "cut32" --timeout=300 2>&1 | tail -3 | tee $OUT/pytest_cut32.txt
XL_TESTING=1 XL_LIBRARY_PATH=$V/libinv32_wpe4.so timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "cut32" --timeout=300 2>&1 | tail -3 | tee -a $OUT/pytest_cut32.txt
echo "== sweeps"
for rnd in 1 2; do
  XL_EXP_INV=3 timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/inv=3      /"
  for inv in 6 7; do
    XL_EXP_INV=$inv timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/inv=$inv wpe3 /"
    XL_TESTING=1 XL_LIBRARY_PATH=$V/libinv32_wpe4.so XL_EXP_INV=$inv timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/inv=$inv wpe4 /"
  done
done | tee $OUT/sweep_inv.txt
