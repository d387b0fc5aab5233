#!/bin/bash
# round 5, session b: float32 matrix-core mix with its A operands staged through LDS (default) against the first build (every lane
# loads its operand straight from the image: -DXLMF_DIRECT) and the staged kernel without its global loads (-DXLMF_EXP_NOLOAD: the
# rate of the matrix pipe + LDS alone; wrong results).
# Usage: gpurun --timeout 1200 -- 'bash tools/gpu_r05b.sh r05b'
TAG=${1:-r05b}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
echo "== pytest (float32 matrix-core mix, staged)"
( time timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=600 -k "f32 or mf32 or config5 or size_rule or other_shapes or other_branch_counts or other_formats" ) > $OUT/pytest_f32mix.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_f32mix.txt
grep -E "passed|failed|exit|real|Error|error" $OUT/pytest_f32mix.txt | tail -8
for lib in default mixf_direct mixf_noload; do
  echo "== $lib"
  if [ $lib = default ]; then unset XL_TESTING XL_LIBRARY_PATH; else export XL_TESTING=1 XL_LIBRARY_PATH=$V/lib$lib.so; fi
  timeout 300 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 --opt mix_kernel=3 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/$lib /" | tee -a $OUT/ab_mix_f32.txt
  timeout 300 python tools/group_sweep.py --shape config5 --clients 256,1024,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/$lib config5 /" | tee -a $OUT/ab_mix_f32.txt
done
unset XL_TESTING XL_LIBRARY_PATH
echo "== staged: passes per workgroup"
for pp in 2 4 8 16; do
  timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 320 --opt mix_kernel=3 --opt mix_passes_per_workgroup=$pp 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/pp=$pp /" | tee -a $OUT/ab_mix_pp.txt
done
echo "== one block per call"
timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 1 --modes optimized --poly3 --blocks 320 --opt mix_kernel=3 2>&1 | grep -v amdgpu.ids | grep optimized | tee -a $OUT/ab_mix_one_block.txt
timeout 300 python tools/group_sweep.py --shape config5 --clients 64,1024 --groups 1 --modes optimized --poly3 --blocks 320 2>&1 | grep -v amdgpu.ids | grep optimized | tee -a $OUT/ab_mix_one_block.txt
