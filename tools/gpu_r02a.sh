#!/bin/bash
# round-2 session a: parity of the refactored engine (groups, merged classes, unlimited classes), first timings.
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6; nproc) > $OUT/env.log 2>&1
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=300 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $OUT/smoke.log
echo "== chain ubench"
timeout 60 ./sdr-server_amd/build/ubench_chain6 2>&1 | tee $OUT/ubench_chain6.txt
echo "== group sweep"
timeout 600 python tools/group_sweep.py --clients 128,256,1024,4096 --groups 1,2,4,8 --modes optimized --m 0 --poly3 2>&1 | grep -v amdgpu.ids | tee $OUT/group_sweep.txt
timeout 300 python tools/group_sweep.py --clients 1024 --groups 4,8 --modes optimized --m 128,256 --poly3 2>&1 | grep -v amdgpu.ids | tee $OUT/group_sweep_m.txt
timeout 300 python tools/group_sweep.py --clients 128,1024 --groups 1,8 --modes native 2>&1 | grep -v amdgpu.ids | tee $OUT/group_sweep_native.txt
