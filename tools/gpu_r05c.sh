#!/bin/bash
# round 5, session c: 16 segments per mix pass (XLP_SEG = 16: all 32 rows of a matrix instruction used; tools/experiments/build_seg16.sh)
# against 14, both matrix-core mix kernels; the chain kernel's clock beside the float32 matrix mix; inverse kernels 3 / 5 alternating.
# Usage: gpurun --timeout 1200 -- 'bash tools/gpu_r05c.sh r05c'
TAG=${1:-r05c}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
echo "== pytest under the 16-segment build (forced polyphase, both matrix-core mixes)"
( time XL_TESTING=1 XL_LIBRARY_PATH=$V/libseg16.so timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -q --timeout=600 -k "(polyphase_forced or group_of_blocks_polyphase or other_branch_counts or tap_scales or config5 or bench_shape_1024_clients_sampled) and not fma and not fused" ) > $OUT/pytest_seg16.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_seg16.txt
grep -E "passed|failed|exit|real|Error|error" $OUT/pytest_seg16.txt | tail -8
for lib in default seg16 default seg16; do
  echo "== $lib"
  if [ $lib = default ]; then unset XL_TESTING XL_LIBRARY_PATH; else export XL_TESTING=1 XL_LIBRARY_PATH=$V/lib$lib.so; fi
  for mk in 1 3; do
    timeout 300 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 --opt mix_kernel=$mk 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/$lib /" | tee -a $OUT/ab_seg16.txt
  done
  timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 1 --modes optimized --poly3 --blocks 320 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/$lib /" | tee -a $OUT/ab_seg16.txt
  timeout 300 python tools/group_sweep.py --shape config5 --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/$lib config5 /" | tee -a $OUT/ab_seg16.txt
done
unset XL_TESTING XL_LIBRARY_PATH
echo "== chain clock beside each mix kernel"
for mk in 1 3 0; do
  XL_EXP_CHAIN_STATS=1 timeout 300 python tools/group_sweep.py --clients 1024,2048 --groups 8 --modes optimized --blocks 640 --opt mix_kernel=$mk 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/mix=$mk /" | tee -a $OUT/chain_clock.txt
done
echo "== inverse kernels alternating"
for rnd in 1 2 3; do for inv in 5 3; do
  timeout 300 python tools/group_sweep.py --clients 2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 --opt inverse_kernel=$inv 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/inv=$inv /" | tee -a $OUT/ab_inverse.txt
done; done
