#!/bin/bash
# round 5, session d: float32 matrix-core mix in Gauss's three-product form (super-passes of 32 segments), 16 segments per pass
# everywhere, the packed-FMA mix kernel gone.  Usage: gpurun --timeout 1200 -- 'bash tools/gpu_r05d.sh r05d'
TAG=${1:-r05d}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest (polyphase path)"
( time timeout 1000 python -m pytest tests/test_batch_gpu.py -m gpu -q --timeout=600 -x -k "polyphase or mix or config5 or size_rule or bench_shape or 2048 or 4096" ) > $OUT/pytest_poly.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_poly.txt
grep -E "passed|failed|exit|real|Error|error" $OUT/pytest_poly.txt | tail -8
for mk in 1 3 1 3; do
  timeout 300 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 --opt mix_kernel=$mk 2>&1 | grep -v amdgpu.ids | grep optimized | tee -a $OUT/ab_mix.txt
done
timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 1 --modes optimized --poly3 --blocks 320 --opt mix_kernel=3 2>&1 | grep -v amdgpu.ids | grep optimized | tee -a $OUT/ab_mix.txt
timeout 300 python tools/group_sweep.py --shape config5 --clients 64,256,1024,4096 --groups 1,8 --modes optimized --poly3 --blocks 320 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/config5 /" | tee -a $OUT/ab_mix.txt
for pp in 2 4 8 16; do
  timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 320 --opt mix_kernel=3 --opt mix_passes_per_workgroup=$pp 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/pp=$pp /" | tee -a $OUT/ab_mix_pp.txt
done
