#!/bin/bash
# tools/gpu_r04r.sh -- round 4, session r: the inverse launch as a PERSISTENT launch (k workgroups per CU walk the tiles, the next
# tile's loads in flight during the transforms): builds with 3 waves per SIMD (160 registers) and 2 (230: addresses hoisted out of
# the tile loop) against the committed one-tile-per-workgroup kernel; parity first, then us per block
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r04r; mkdir -p $OUT
export TMPDIR=/tmp XL_TESTING=1
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
{
for cfg in "inv_w3 3" "inv_h2 2"; do set -- $cfg
  echo "== parity: $1, inverse_persistent=$2"
  XL_LIBRARY_PATH=$V/lib$1.so XL_EXP_INV_PERSIST=$2 timeout 300 python -m pytest tests/test_batch_gpu.py -q -x -k "group_2048_clients_all or polyphase_matches or group_of_blocks_polyphase or churn" 2>&1 | tail -3
done
for cfg in "inv_head 0" "inv_w3 0" "inv_w3 3" "inv_w3 4" "inv_h2 2" "inv_h2 3" "inv_head 0"; do set -- $cfg
  echo "== $1, inverse_persistent=$2"
  XL_LIBRARY_PATH=$V/lib$1.so timeout 200 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --blocks 160 --poly3 --opt inverse_persistent=$2 2>&1 | grep "^optimized"
  XL_LIBRARY_PATH=$V/lib$1.so timeout 100 python tools/group_sweep.py --clients 1024 --groups 1 --blocks 320 --poly3 --opt inverse_persistent=$2 2>&1 | grep "^optimized"
done
} 2>&1 | tee $OUT/inverse_persistent.txt | cut -c1-200
