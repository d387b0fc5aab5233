#!/bin/bash
# tools/r06_m64.sh <outdir> -- a `custom:` step of tools/gpu_r06.sh: 64-point transforms (option polyphase_m = 64; round 6, session 2): parity of the
# forced-path tests, then which transform length is fastest where -- config 5 (cf32, D = 100, 3 taps per branch) over client counts and
# blocks per call, and cu8 shapes of other branch counts / taps per branch
OUT=$1
timeout 1500 python -m pytest tests/test_batch_gpu.py -m gpu -q --timeout=900 -k "M64" > $OUT/pytest_m64.txt 2>&1
tail -3 $OUT/pytest_m64.txt
for m in 64 128 256; do
  timeout 300 python tools/group_sweep.py --shape config5 --clients 128,256,512,768,1024,2048,4096 --groups 8 --modes optimized --blocks 320 --m $m 2>&1 | grep optimized | sed "s/^/config5 /"
  timeout 300 python tools/group_sweep.py --shape config5 --clients 256,1024,4096 --groups 1,2 --modes optimized --blocks 320 --m $m 2>&1 | grep optimized | sed "s/^/config5 /"
  for D in 64 72 100; do
    for rate in 1 2 3; do
      timeout 300 python tools/group_sweep.py --decimations $D --rate $rate --clients 1024,4096 --groups 8 --modes optimized --blocks 320 --m $m 2>&1 | grep optimized | sed "s/^/cu8-D$D-rate$rate /"
    done
  done
done | tee $OUT/m64_sweep.txt
