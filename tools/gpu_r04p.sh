#!/bin/bash
# tools/gpu_r04p.sh -- round 4, session p: (1) the "expected_clients" test, (2) decimations 42..64 on the matrix-core mix (49..64: the
# instantiations with spilled registers) against the packed-FMA mix, (3) one block per call at 1024 clients: the inverse launch's
# variants, the transform length, temporal Y stores, and the per-workgroup phases of the inverse launch (tuning build of xl_batch.cpp)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r04p; mkdir -p $OUT
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
timeout 240 python -m pytest tests/test_batch_gpu.py -q -x -k "expected_clients" > $OUT/pytest_expected.txt 2>&1; tail -3 $OUT/pytest_expected.txt
{
for mk in 1 0; do echo "== mix_kernel=$mk"; timeout 300 python tools/group_sweep.py --clients 1024,2048 --groups 8 --decimations 42,48,50,56,64 --blocks 160 --poly3 --opt mix_kernel=$mk 2>&1 | grep -v "^mode"; done
} > $OUT/d_sweep.txt 2>&1
{
for ik in 3 0 4 1 2; do echo "== inverse_kernel=$ik"; timeout 120 python tools/group_sweep.py --clients 1024 --groups 1 --blocks 320 --poly3 --opt inverse_kernel=$ik 2>&1 | grep -v "^mode"; done
echo "== polyphase_m=256"; timeout 120 python tools/group_sweep.py --clients 1024 --groups 1 --blocks 320 --poly3 --m 256 2>&1 | grep -v "^mode"
echo "== Y stores temporal"; XL_TESTING=1 XL_LIBRARY_PATH=$V/libytemporal.so timeout 120 python tools/group_sweep.py --clients 1024 --groups 1 --blocks 320 --poly3 2>&1 | grep -v "^mode"
echo "== default again"; timeout 120 python tools/group_sweep.py --clients 1024 --groups 1 --blocks 320 --poly3 2>&1 | grep -v "^mode"
} > $OUT/one_block_variants.txt 2>&1
{
for G in 1 8; do
  echo "== inverse launch, per-workgroup phases (first wave of each workgroup), 1024 clients, $G block(s) per call"
  XL_TESTING=1 XL_LIBRARY_PATH=$V/libtuning.so XL_EXP_POLY_TRACE=/tmp/tr$G.bin XL_EXP_POLY_TRACE_INV=1 timeout 120 python tools/group_sweep.py --clients 1024 --groups $G --blocks 16 2>&1 | grep -v "^mode"
  python tools/inv_trace.py /tmp/tr$G.bin $((G == 1 ? 864 : 6000))
done
} > $OUT/inverse_trace.txt 2>&1
cat $OUT/d_sweep.txt $OUT/one_block_variants.txt $OUT/inverse_trace.txt | cut -c1-230
