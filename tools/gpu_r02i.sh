#!/bin/bash
TAG=${1:-r02i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for ev in 0 1; do
  if [ $ev = 1 ]; then export XL_EXP_EVDONE=1; else unset XL_EXP_EVDONE; fi
  XL_EXP_NCO_SIDE=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_ev$ev -o t -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 8 --modes optimized --blocks 96 > $OUT/t_ev$ev.log 2>&1
  f=$(find $OUT/t_ev$ev -name "*kernel_trace.csv" | head -1)
  echo "== side 0, ev_done record $ev"; python3 $GRAFT_REPO_ROOT/tools/timeline.py $f 8 2 | grep -E "fwd|call period"
done
unset XL_EXP_EVDONE
cd $GRAFT_REPO_ROOT
echo "== wall-clock sweep (no profiler)"
for side in 1 0; do
  XL_EXP_NCO_SIDE=$side timeout 600 python tools/group_sweep.py --clients 128,512,1024,2048,4096 --groups 8 --modes optimized --blocks 640 2>&1 | grep -v amdgpu.ids | sed "s/^/side=$side /" | tee -a $OUT/sweep.txt
done
for side in 1 0; do
  XL_EXP_NCO_SIDE=$side timeout 600 python tools/group_sweep.py --clients 1024 --groups 1,2,4 --modes optimized --blocks 640 2>&1 | grep -v amdgpu.ids | sed "s/^/side=$side /" | tee -a $OUT/sweep.txt
done
