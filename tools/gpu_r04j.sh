#!/bin/bash
# round 4, session j: the role on six SCALAR instructions (inline asm: the compiler re-packs C): soak x3 + diagnostics
TAG=${1:-r04j}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3; do SOAK_ONLY_PAIR=1 SOAK_CALLS=2000 timeout 120 python tools/experiments/dbg_soak.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/dbg_soak_pair.txt; done
timeout 600 python -m pytest tests/test_batch_gpu.py tests/test_gpu_parity.py -m gpu -q -x --timeout=400 -k "soak or role or native or drift or x86" 2>&1 | tail -4 | tee $OUT/pytest_role.txt
