#!/bin/bash
# tools/gpu_r04x.sh -- round 4, session x: why do the inverse launch's stores run at 3.0 TB/s when a store-only kernel of the same pattern
# reaches 4.4-5.0?  (a) the microbenchmark with the engine's row pitch, (b) the launch's store-only build without its column records /
# phase-table entries (no dependent loads ahead of the stores)
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r04x; mkdir -p $OUT
export TMPDIR=/tmp XL_TESTING=1
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
{
timeout 120 ./sdr-server_amd/build/ubench_store_pattern 4096 35840 199936 | head -4
for v in inv8_wr inv8_wr_nometa inv8_wr inv8_wr_nometa; do
  echo "== $v"
  XL_LIBRARY_PATH=$V/lib$v.so timeout 200 python tools/group_sweep.py --clients 4096 --groups 8 --blocks 160 --poly3 --opt inverse_kernel=5 2>&1 | grep "^optimized"
done
} 2>&1 | tee $OUT/store_rate_hunt.txt | cut -c1-200
