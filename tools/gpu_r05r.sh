#!/bin/bash
# round 5, session r: the mix launches with XCD x taking M/8 ADJACENT bins of every column group (new) against bins = x mod 8 (round 3's
# placement; variant libraries built with -DXLP_MIX_BINS_STRIDED), alternating; parity of the forced-path tests first.
TAG=${1:-r05r}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
echo "== parity (forced polyphase tests, group tests, config 5)"
timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "polyphase or config5 or group_of_blocks or matrix_core" --timeout=300 2>&1 | tail -3 | tee $OUT/pytest_poly.txt
echo "== sweeps"
for rnd in 1 2; do
  timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/adjacent /"
  XL_TESTING=1 XL_LIBRARY_PATH=$V/libmix_strided.so timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/strided  /"
  XL_EXP_MIX=3 timeout 200 python tools/group_sweep.py --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/adjacent f32 /"
  XL_EXP_MIX=3 XL_TESTING=1 XL_LIBRARY_PATH=$V/libmixf_strided.so timeout 200 python tools/group_sweep.py --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/strided  f32 /"
  timeout 200 python tools/group_sweep.py --shape config5 --clients 1024 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/adjacent config5 /"
  XL_TESTING=1 XL_LIBRARY_PATH=$V/libmixf_strided.so timeout 200 python tools/group_sweep.py --shape config5 --clients 1024 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/strided  config5 /"
done | tee $OUT/sweep_mix_place.txt
for rnd in 1 2; do
  timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 1 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/adjacent /"
  XL_TESTING=1 XL_LIBRARY_PATH=$V/libmix_strided.so timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 1 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/strided  /"
done | tee $OUT/sweep_mix_place_g1.txt
