#!/bin/bash
# forward launch: branch-major over the XCDs (default) against pass-major (variant library), alternating; parity subset first
OUT=$1; V=sdr-server_amd/build/variants
timeout 1500 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=900 -k "config5 or adversarial or polyphase_forced or group_of_blocks_polyphase or bench_shape_1024_clients_all or staggered or churn" > $OUT/pytest_fwd.txt 2>&1
tail -3 $OUT/pytest_fwd.txt
for rep in 1 2 3; do
  timeout 200 python tools/group_sweep.py --shape config5 --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 1600 2>&1 | grep optimized | sed "s/^/branch-major /"
  XL_LIBRARY_PATH=$V/libfwd_PASSMAJOR.so timeout 200 python tools/group_sweep.py --shape config5 --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 1600 2>&1 | grep optimized | sed "s/^/pass-major   /"
  timeout 200 python tools/group_sweep.py --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 1600 2>&1 | grep optimized | sed "s/^/branch-major /"
  XL_LIBRARY_PATH=$V/libfwd_PASSMAJOR.so timeout 200 python tools/group_sweep.py --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 1600 2>&1 | grep optimized | sed "s/^/pass-major   /"
done | tee $OUT/forward_xcd_ab.txt
