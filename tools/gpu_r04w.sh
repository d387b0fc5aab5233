#!/bin/bash
# tools/gpu_r04w.sh -- round 4, session w: valid outputs per segment rounded down to a multiple of 16 (V = 112 instead of 116 at 128 points:
# every (segment, client) output piece starts and ends on a 128-byte line, every phase walk starts on a table entry; 3.5 % more segments)
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r04w; mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== parity with XL_EXP_V16=1"
XL_EXP_V16=1 timeout 400 python -m pytest tests/test_batch_gpu.py -q -k "group_2048_clients_all or churn or staggered or group_of_blocks_polyphase or ragged" 2>&1 | tail -4
for v in 0 1 0 1; do echo "== V16=$v"
  if [ $v = 1 ]; then export XL_EXP_V16=1; else unset XL_EXP_V16; fi
  timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --blocks 160 --poly3 2>&1 | grep "^optimized"
  timeout 100 python tools/group_sweep.py --clients 1024 --groups 1 --blocks 320 --poly3 2>&1 | grep "^optimized"
done
} 2>&1 | tee $OUT/v16.txt | cut -c1-200
