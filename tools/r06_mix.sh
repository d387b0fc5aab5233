#!/bin/bash
# tools/r06_mix.sh <outdir> -- a `custom:` step of tools/gpu_r06.sh: the wide two-half mix (xl_mixh2.hip) and the cf32 segment scales:
# parity subset, BASELINE config 5 timing (two-half default against option mix_kernel = 3) over client counts and blocks per call, and --
# when a -DXL_TUNING library was built (sdr-server_amd/build/variants/libtune.so) -- the mix launch's own timeline (tools/r06_trace.py)
OUT=$1; V=sdr-server_amd/build/variants
timeout 1500 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=900 -k "config5 or adversarial or forced_other_shapes or size_rule or forced_other_formats or plain_process or other_input_formats or other_branch_counts or tap_scales or group_of_blocks_polyphase" > $OUT/pytest_mix.txt 2>&1
tail -5 $OUT/pytest_mix.txt
for rep in 1 2; do
  timeout 200 python tools/group_sweep.py --shape config5 --clients 256,512,1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/two-half      /"
  timeout 200 python tools/group_sweep.py --shape config5 --clients 1024 --groups 8 --modes optimized --poly3 --blocks 320 --opt mix_kernel=3 2>&1 | grep optimized | sed "s/^/float32       /"
  timeout 200 python tools/group_sweep.py --shape config5 --clients 1024 --groups 1,2,4 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/two-half      /"
done | tee $OUT/config5_mix_ab.txt
if [ -f $V/libtune.so ]; then
  XL_LIBRARY_PATH=$V/libtune.so XL_EXP_POLY_TRACE=$OUT/mix_trace.bin timeout 300 python tools/group_sweep.py --shape config5 --clients 1024 --groups 8 --modes optimized --blocks 64 2>&1 | grep optimized
  python tools/r06_trace.py $OUT/mix_trace.bin 4096 | tee $OUT/mix_trace_1024.txt
  rm -f $OUT/mix_trace.bin
fi
