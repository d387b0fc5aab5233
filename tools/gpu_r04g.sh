#!/bin/bash
# round 4, session g: why are pipelined one-block calls slower?  Kernel timelines (rocprofv3 --kernel-trace) of one block per call at
# 1024 clients with and without "pipeline_calls"; the feed-timing test again.
TAG=${1:-r04g}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=200 -k "feed_timing" 2>&1 | tail -3
cd /tmp
for p in 1 0; do
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace$p -o t -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 1 --blocks 80 --opt pipeline_calls=$p > $OUT/trace$p.log 2>&1
  f=$(find $OUT/trace$p -name "*kernel_trace.csv" | head -1)
  echo "== pipeline_calls=$p"; grep optimized $OUT/trace$p.log
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $f 40 6 | tee $OUT/timeline_pipe$p.txt
done
find $OUT -name "*.csv" -delete
