#!/bin/bash
# round 5, session a: the mix launch on the matrix cores with float32 operands (mix_kernel 3) -- parity of the new tests, then A/B
# against the two-half float16 mix (1) and the packed-FMA mix (0), passes per workgroup, BASELINE config 5.
# Usage: gpurun --timeout 1500 -- 'bash tools/gpu_r05a.sh r05a'
TAG=${1:-r05a}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo) > $OUT/env.txt 2>&1
echo "== pytest (float32 matrix-core mix)"
( time timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=600 -k "f32 or mf32 or config5 or size_rule or other_shapes or other_branch_counts or other_formats or fixture_shape or ragged_and_join" ) > $OUT/pytest_f32mix.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_f32mix.txt
grep -E "passed|failed|exit|real|Error|error" $OUT/pytest_f32mix.txt | tail -8
echo "== A/B mix kernels, 8 blocks per call"
for mk in 1 3 0; do
  timeout 300 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 --opt mix_kernel=$mk 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_mix.txt
done
echo "== float32 matrix mix: passes per workgroup"
for pp in 2 4 8 16; do
  timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 320 --opt mix_kernel=3 --opt mix_passes_per_workgroup=$pp 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/pp=$pp /" | tee -a $OUT/ab_mix_pp.txt
done
echo "== one block per call"
for mk in 1 3; do
  timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 1 --modes optimized --poly3 --blocks 320 --opt mix_kernel=$mk 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/mix=$mk /" | tee -a $OUT/ab_mix_one_block.txt
done
echo "== config 5 (cf32 10 Msps, D=100, 257 taps)"
timeout 300 python tools/group_sweep.py --shape config5 --clients 64,256,1024,4096 --groups 1,8 --modes optimized --poly3 --blocks 320 2>&1 | grep -v amdgpu.ids | tee $OUT/config5.txt
timeout 300 python tools/group_sweep.py --shape config5 --clients 64,256,1024 --groups 8 --modes optimized --blocks 320 --opt polyphase=0 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/direct /" | tee -a $OUT/config5.txt
timeout 300 python tools/group_sweep.py --shape config5 --clients 256,1024 --groups 8 --modes optimized --poly3 --blocks 320 --opt mix_kernel=0 --opt polyphase=1 2>&1 | grep -v amdgpu.ids | grep optimized | sed "s/^/fma-mix /" | tee -a $OUT/config5.txt
timeout 300 python tools/group_sweep.py --shape config5 --clients 1024 --groups 8 --modes native --blocks 160 2>&1 | grep -v amdgpu.ids | grep native | tee -a $OUT/config5.txt
echo "== rocprofv3 kernel stats, config 5 at 1024 clients"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof5 -o c5 -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --shape config5 --clients 1024 --groups 8 --modes optimized --blocks 640 > $OUT/prof5.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof5 -name "*kernel_stats*.csv" | head -1); do head -8 $f; cp $f $OUT/rocprofv3_kernel_stats_config5_1024clients.csv; done
rm -rf $OUT/prof5
echo "== rocprofv3 kernel stats, all-float32 server shape at 1024 clients"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof3 -o f32 -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 8 --modes optimized --blocks 640 --opt mix_kernel=3 > $OUT/prof3.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof3 -name "*kernel_stats*.csv" | head -1); do head -8 $f; cp $f $OUT/rocprofv3_kernel_stats_f32mix_1024clients.csv; done
rm -rf $OUT/prof3
