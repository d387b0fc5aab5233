#!/bin/bash
# round 4, session a: first run of the fused mix + inverse launch -- parity of the fused variants, then A/B timing against the
# three-launch path (same box): clients 1024 / 2048 / 4096, 8 blocks and 1 block per call, per-launch times.
TAG=r04a; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc) > $OUT/env.txt 2>&1
echo "== pytest fused"
timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=600 -k "fused or role_phases" > $OUT/pytest_fused.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_fused.txt; tail -15 $OUT/pytest_fused.txt
echo "== sweep mfma (three launches)"
timeout 400 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8,1 --poly3 --blocks 160 --opt mix_kernel=1 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_mfma.txt
echo "== sweep fused"
timeout 400 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8,1 --poly3 --blocks 160 --opt mix_kernel=2 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_fused.txt
echo "== sweep fused, split forced on / off at 8 blocks"
timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 8 --poly3 --blocks 160 --opt mix_kernel=2 --opt fused_split=1 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_fused_split1.txt
timeout 300 python tools/group_sweep.py --clients 1024 --groups 1 --poly3 --blocks 160 --opt mix_kernel=2 --opt fused_split=0 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_fused_split0.txt
