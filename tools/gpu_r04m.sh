#!/bin/bash
# round 4, session m: 48-bit Y, rows padded to whole lines: quick parity + A/B
TAG=${1:-r04m}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=300 -k "bench_shape_1024_clients_all and (optimized)" 2>&1 | tail -2
for y in 0 1 0 1; do
  echo "== y_format=$y"
  timeout 300 python tools/group_sweep.py --clients 2048,4096 --groups 8 --poly3 --blocks 240 --opt y_format=$y 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep_y$y.txt
done
