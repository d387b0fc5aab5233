#!/bin/bash
# round 5, session ai: the float32 mix with (1) the first pass's products running as the operands arrive, (2) workgroup barriers that do not fence global memory, (3) unconditional row loads
# -- against the previous commit's kernel (variant library), alternating; parity first
TAG=${1:-r05ai}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x -k "config5 or f32 or other_formats or other_shapes" --timeout=300 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2 | tee $OUT/pytest.txt
for rnd in 1 2; do
  timeout 200 python tools/group_sweep.py --shape config5 --clients 512,1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/new  /"
  XL_TESTING=1 XL_LIBRARY_PATH=$V/libmixf_prev.so timeout 200 python tools/group_sweep.py --shape config5 --clients 512,1024,2048,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/prev /"
  XL_EXP_MIX=3 timeout 200 python tools/group_sweep.py --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/new  f32 /"
  XL_EXP_MIX=3 XL_TESTING=1 XL_LIBRARY_PATH=$V/libmixf_prev.so timeout 200 python tools/group_sweep.py --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/prev f32 /"
done | tee $OUT/sweep_mixf.txt
