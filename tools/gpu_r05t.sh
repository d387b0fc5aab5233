#!/bin/bash
# round 5, session t: the two-half mix launch with a LOADER wave (5 waves per workgroup: the product waves never wait for their stores)
# against the 4-wave form (variant library built from the previous commit's xl_polyphase.hip), alternating; parity first.
TAG=${1:-r05t}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
echo "== parity (forced polyphase tests, group tests, 1024 / 2048 / 4096 populations)"
timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "polyphase or group_of_blocks or matrix_core or 1024_clients or 2048_clients or 4096_clients or size_rule" --timeout=300 2>&1 | tail -3 | tee $OUT/pytest_poly.txt
echo "== sweeps"
for rnd in 1 2; do
  timeout 200 python tools/group_sweep.py --clients 128,1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/loader /"
  XL_TESTING=1 XL_LIBRARY_PATH=$V/libmix4waves.so timeout 200 python tools/group_sweep.py --clients 128,1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/4waves /"
done | tee $OUT/sweep_mix_loader.txt
for rnd in 1 2; do
  timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 1 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/loader /"
  XL_TESTING=1 XL_LIBRARY_PATH=$V/libmix4waves.so timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 1 --modes optimized --blocks 320 2>&1 | grep optimized | sed "s/^/4waves /"
done | tee $OUT/sweep_mix_loader_g1.txt
