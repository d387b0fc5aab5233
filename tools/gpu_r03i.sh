#!/bin/bash
# round 3, GPU session i: the mix launch on the matrix cores (xlp_mix_mfma_kernel) -- parity first, then A/B against the FMA kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03i; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -x -q -k "polyphase or group_of_blocks or bench_shape" ) > $OUT/pytest_poly.log 2>&1
tail -15 $OUT/pytest_poly.log
for mix in 0 1; do
  XL_EXP_MIX=$mix timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 8 --poly3 --blocks 640 > $OUT/sweep_mix$mix.txt 2>&1
  echo "== mix $mix"; grep -v amdgpu $OUT/sweep_mix$mix.txt
done
for pp in 2 8 16; do
  XL_EXP_MIX=1 XL_EXP_MIX_PP=$pp timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 8 --poly3 --blocks 640 > $OUT/sweep_mix1_pp$pp.txt 2>&1
  echo "== mix 1 pp $pp"; grep -v amdgpu $OUT/sweep_mix1_pp$pp.txt
done
