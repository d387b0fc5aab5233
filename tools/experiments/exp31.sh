#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/s31; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for n in 1024 4096; do
XL_EXP_POLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$n -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients $n --rates 5 --modes optimized --steps 50 > $OUT/prof$n.log 2>&1
f=$(find $OUT/prof$n -name "*kernel_stats.csv" | head -1); echo "== $n clients"; tail -1 $OUT/prof$n.log; column -s, -t "$f" | cut -c1-150 | head -12
done
