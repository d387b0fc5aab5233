#!/bin/bash
# tools/gpu_r04z.sh -- round 4, session z: the 8-lane inverse launch with (a) the column record requested ahead of the tile and the table entry
# ahead of the twiddle table's fill, (b) every load waited for once, before the first predicated block (no s_waitcnt vmcnt(0) between the
# epilogue's stores any more) -- against the committed build of the same kernel, same box
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r04z3; mkdir -p $OUT
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
{
echo "== parity (new build)"
timeout 400 python -m pytest tests/test_batch_gpu.py -q -x -k "8-lane or group_2048_clients_all or group_4096_clients_sampled or churn or staggered" 2>&1 | tail -2
for v in inv8_head new inv8_head new; do
  if [ $v = new ]; then unset XL_TESTING XL_LIBRARY_PATH; else export XL_TESTING=1 XL_LIBRARY_PATH=$V/lib$v.so; fi
  echo "== $v"
  timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --blocks 160 --poly3 2>&1 | grep "^optimized"
  timeout 100 python tools/group_sweep.py --clients 1024 --groups 1 --blocks 320 --poly3 2>&1 | grep "^optimized"
done
} 2>&1 | tee $OUT/inverse8_waits.txt | cut -c1-200
