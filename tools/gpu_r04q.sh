#!/bin/bash
# tools/gpu_r04q.sh -- round 4, session q: one block per call, 1024 clients: kernel timeline (rocprofv3 --kernel-trace) of the shipped
# build and of a build whose inverse launch writes the outputs with streaming stores; throughput of both
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r04q; mkdir -p $OUT
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
cd /tmp
for v in shipped outnt; do
  if [ $v = shipped ]; then unset XL_TESTING XL_LIBRARY_PATH; else export XL_TESTING=1 XL_LIBRARY_PATH=$V/lib$v.so; fi
  echo "== $v"
  for c in 1024 2048; do timeout 120 python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients $c --groups 1,8 --blocks 320 2>&1 | grep "^optimized"; done
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$v -o p -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 1 --blocks 128 > $OUT/kt_$v.log 2>&1
  f=$(find $OUT/kt_$v -name '*kernel_trace.csv' | head -1)
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $f 60 3
  rm -rf $OUT/kt_$v
done 2>&1 | tee $OUT/one_block_outnt.txt | cut -c1-200
