#!/bin/bash
# tools/r06_fwdrole.sh <outdir> -- a `custom:` step of tools/gpu_r06.sh: the forward transforms as a role of the two-half mix launch
# (xl_fwd_role.h): parity subset first (every polyphase test), then A/B against a launch of their own (XL_EXP_FWD_IN_MIX=0), alternating
OUT=$1
timeout 1500 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=900 -k "config5 or adversarial or polyphase or size_rule or forced or plain_process or other_input_formats or other_branch_counts or tap_scales or bench_shape or staggered or churn or mixed_rates" > $OUT/pytest_fwdrole.txt 2>&1
tail -5 $OUT/pytest_fwdrole.txt
for rep in 1 2 3; do
  for shape in config5 server; do
    timeout 200 python tools/group_sweep.py --shape $shape --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 640 2>&1 | grep optimized | sed "s/^/in-mix   $shape /"
    XL_EXP_FWD_IN_MIX=0 timeout 200 python tools/group_sweep.py --shape $shape --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 640 2>&1 | grep optimized | sed "s/^/separate $shape /"
  done
  timeout 200 python tools/group_sweep.py --clients 128,1024 --groups 1,2 --modes optimized --blocks 640 2>&1 | grep optimized | sed "s/^/in-mix   server /"
  XL_EXP_FWD_IN_MIX=0 timeout 200 python tools/group_sweep.py --clients 128,1024 --groups 1,2 --modes optimized --blocks 640 2>&1 | grep optimized | sed "s/^/separate server /"
done | tee $OUT/forward_in_mix_ab.txt
V=sdr-server_amd/build/variants
if [ -f $V/libtune.so ]; then
  for shape in config5 server; do
    XL_LIBRARY_PATH=$V/libtune.so XL_EXP_POLY_TRACE=$OUT/mix_trace.bin timeout 300 python tools/group_sweep.py --shape $shape --clients 1024 --groups 8 --modes optimized --blocks 64 2>&1 | grep optimized
    echo "# $shape 1024 clients x 8 blocks: the mix launch with the forward role"; python tools/r06_trace.py $OUT/mix_trace.bin 4096
    XL_EXP_FWD_IN_MIX=0 XL_LIBRARY_PATH=$V/libtune.so XL_EXP_POLY_TRACE=$OUT/mix_trace.bin timeout 300 python tools/group_sweep.py --shape $shape --clients 1024 --groups 8 --modes optimized --blocks 64 2>&1 | grep optimized
    echo "# $shape 1024 clients x 8 blocks: the mix launch alone"; python tools/r06_trace.py $OUT/mix_trace.bin 4096
  done | tee $OUT/mix_trace_fwdrole.txt
  rm -f $OUT/mix_trace.bin
fi
