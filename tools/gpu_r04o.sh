#!/bin/bash
# round 4, session o: the stand-alone table kernel on an exclusive SIMD (packed step): drop-in parity + latency
TAG=${1:-r04o}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_c_dropin.py -m gpu -q -x --timeout=300 2>&1 | grep -E "passed|failed" | tail -2
timeout 200 python tools/measure_dropin.py 2>&1 | grep -v amdgpu.ids | tee $OUT/dropin.txt | tail -25
