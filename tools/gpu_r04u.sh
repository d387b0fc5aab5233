#!/bin/bash
# tools/gpu_r04u.sh -- round 4, session u: what bounds the inverse launch's traffic -- reads only, writes only, and the tiles handed out
# segment-fastest (a client row's neighbouring pieces written at about the same time); 8-lane kernel, us per block
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r04u; mkdir -p $OUT
export TMPDIR=/tmp XL_TESTING=1
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
{
for v in inv8_copy inv8_rd inv8_wr inv8_copy_sfast inv8_sfast; do
  echo "== $v"
  XL_LIBRARY_PATH=$V/lib$v.so timeout 200 python tools/group_sweep.py --clients 1024,4096 --groups 8 --blocks 160 --poly3 --opt inverse_kernel=5 2>&1 | grep "^optimized"
done
echo "== parity of the segment-fastest order"
XL_LIBRARY_PATH=$V/libinv8_sfast.so XL_EXP_INV=5 timeout 300 python -m pytest tests/test_batch_gpu.py -q -x -k "group_2048_clients_all or churn" 2>&1 | tail -2
} 2>&1 | tee $OUT/inverse_traffic_split.txt | cut -c1-200
