#!/bin/bash
# tools/r06_fwd4.sh <outdir> -- a `custom:` step of tools/gpu_r06.sh: the forward launch with groups of adjacent branches per workgroup (one
# load per NB samples) against one branch per workgroup (variant library = the previous build), alternating; parity subset first; then
# the launch's own timeline (-DXL_TUNING library)
OUT=$1; V=sdr-server_amd/build/variants
timeout 1500 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=900 -k "config5 or adversarial or polyphase or size_rule or forced or plain_process or other_input_formats or other_branch_counts or tap_scales or bench_shape or staggered or churn or mixed_rates or ragged or late or join" > $OUT/pytest_fwd4.txt 2>&1
tail -5 $OUT/pytest_fwd4.txt
for rep in 1 2; do
  for shape in config5 server; do
    timeout 200 python tools/group_sweep.py --shape $shape --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 640 2>&1 | grep optimized | sed "s/^/groups-of-4   $shape /"
    XL_LIBRARY_PATH=$V/libfwd_old.so timeout 200 python tools/group_sweep.py --shape $shape --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 640 2>&1 | grep optimized | sed "s/^/one-branch    $shape /"
  done
  timeout 200 python tools/group_sweep.py --shape config5 --clients 128,1024 --groups 1,2 --modes optimized --poly3 --blocks 640 2>&1 | grep optimized | sed "s/^/groups-of-4   server /"
  XL_LIBRARY_PATH=$V/libfwd_old.so timeout 200 python tools/group_sweep.py --shape config5 --clients 128,1024 --groups 1,2 --modes optimized --poly3 --blocks 640 2>&1 | grep optimized | sed "s/^/one-branch    server /"
done | tee $OUT/forward_groups_ab.txt
for shape in config5 server; do
  XL_LIBRARY_PATH=$V/libtune.so XL_EXP_POLY_TRACE=$OUT/fwd_trace_$shape.bin XL_EXP_POLY_TRACE_FWD=1 timeout 300 python tools/group_sweep.py --shape $shape --clients 1024 --groups 8 --modes optimized --blocks 64 2>&1 | grep optimized
  echo "# $shape, 1024 clients x 8 blocks"; python tools/fwd_trace.py $OUT/fwd_trace_$shape.bin 700
done | tee $OUT/forward_trace_groups.txt
rm -f $OUT/fwd_trace_*.bin
