#!/bin/bash
# two-half mix for cf32 / D <= 112 classes: parity subset + config 5 timing, default (two-half) against mix_kernel = 3 (float32 operands)
OUT=$1
timeout 1500 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=900 -k "config5 or adversarial or forced_other_shapes or size_rule or forced_other_formats or plain_process or other_input_formats" > $OUT/pytest_mix.txt 2>&1
tail -12 $OUT/pytest_mix.txt
for rep in 1 2; do
  timeout 200 python tools/group_sweep.py --shape config5 --clients 256,1024,2048 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/two-half /"
  timeout 200 python tools/group_sweep.py --shape config5 --clients 256,1024,2048 --groups 8 --modes optimized --poly3 --blocks 320 --opt mix_kernel=3 2>&1 | grep optimized | sed "s/^/float32  /"
done | tee $OUT/config5_mix_ab.txt
