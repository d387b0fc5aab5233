#!/bin/bash
# round 3, GPU session j: matrix-core mix, 4-waves-per-SIMD build: role regression test, polyphase parity, A/B sweeps
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03j; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do env AB_CALLS=120 timeout 250 python tools/experiments/dbg_phase_ab.py 2>&1 | grep -v "amdgpu\|^clients" | cut -c1-150 | tail -3; done
( time timeout 1200 python -m pytest tests/test_batch_gpu.py -m gpu -x -q -k "polyphase or group_of_blocks or bench_shape or matrix_core or staggered" ) > $OUT/pytest_poly.log 2>&1
tail -5 $OUT/pytest_poly.log
XL_EXP_MIX=0 timeout 300 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --poly3 --blocks 640 > $OUT/sweep_mix0.txt 2>&1
echo "== mix 0"; grep -v amdgpu $OUT/sweep_mix0.txt
for pp in 4 8 16; do
  XL_EXP_MIX=1 XL_EXP_MIX_PP=$pp timeout 300 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --poly3 --blocks 640 > $OUT/sweep_mix1_pp$pp.txt 2>&1
  echo "== mix 1 pp $pp"; grep -v "amdgpu\|^mode" $OUT/sweep_mix1_pp$pp.txt
done
XL_EXP_MIX=1 timeout 300 python tools/group_sweep.py --clients 128,1024,4096 --groups 1,16 --poly3 --blocks 640 > $OUT/sweep_mix1_g.txt 2>&1
echo "== mix 1 G 1,16"; grep -v "amdgpu" $OUT/sweep_mix1_g.txt
