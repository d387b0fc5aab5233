#!/bin/bash
# round 4, session d: as c after the index fix, guarded: nothing runs after a failed parity pass, every step has its own short timeout
TAG=${1:-r04d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest fused (quick subset first)"
timeout 240 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=120 -k "fused and (ragged or group_of_blocks)" > $OUT/pytest_quick.txt 2>&1
rc=$?; tail -4 $OUT/pytest_quick.txt; if [ $rc -ne 0 ]; then echo "quick parity failed: stop"; tail -40 $OUT/pytest_quick.txt; exit 1; fi
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=300 -k "fused or role_phases" > $OUT/pytest_fused.txt 2>&1
rc=$?; echo "pytest exit $rc" >> $OUT/pytest_fused.txt; tail -5 $OUT/pytest_fused.txt
echo "== sweep fused"
timeout 300 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8,1 --poly3 --blocks 160 --opt mix_kernel=2 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_fused.txt
echo "== sweep mfma (three launches)"
timeout 300 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8,1 --poly3 --blocks 160 --opt mix_kernel=1 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_mfma.txt
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
for v in f_noepi f_noloads f_sameops; do
  echo "== variant $v"
  XL_LIBRARY_PATH=$V/lib$v.so timeout 120 python tools/group_sweep.py --clients 4096 --groups 8 --poly3 --blocks 160 --opt mix_kernel=2 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_$v.txt
done
