#!/bin/bash
TAG=${1:-r02k}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== parity subset"
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q --timeout=300 -k "group or side_stream or multi or stagger or feeder" 2>&1 | tail -3
echo "== wall-clock sweep: engine stream (CU mask) vs torch stream"
for es in 1 0; do
  timeout 600 python tools/group_sweep.py --clients 128,512,1024,2048 --groups 8 --modes optimized --blocks 640 --engine-stream $es 2>&1 | grep -v amdgpu.ids | sed "s/^/engine_stream=$es /" | tee -a $OUT/sweep.txt
done
timeout 300 python tools/group_sweep.py --clients 1024 --groups 8 --modes native --blocks 320 --engine-stream 1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 8 --modes optimized --blocks 96 > $OUT/t.log 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/timeline.py $f 8 3 | tee $OUT/timeline_1024_masked.txt
