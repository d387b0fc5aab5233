#!/bin/bash
# round-2 session d: kernel timelines (side-stream chain on / off) at 128 and 1024 clients, 8 blocks per call
TAG=${1:-r02d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for n in 128 1024; do for side in 1 0; do
  XL_EXP_NCO_SIDE=$side timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_${n}_$side -o t -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients $n --groups 8 --modes optimized --blocks 96 > $OUT/t_${n}_$side.log 2>&1
  f=$(find $OUT/t_${n}_$side -name "*kernel_trace.csv" | head -1)
  echo "== clients $n side $side"; grep optimized $OUT/t_${n}_$side.log
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $f 8 3 | tee $OUT/timeline_${n}_$side.txt
done; done
