#!/bin/bash
# round 4, session c: the fused launch, second shape (8 x 16 tiles, two workgroups per CU): parity, A/B against the three-launch
# path on one box, time split by variants, counters.
TAG=${1:-r04c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest fused"
timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=600 -k "fused or role_phases" > $OUT/pytest_fused.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_fused.txt; tail -5 $OUT/pytest_fused.txt
echo "== sweep mfma (three launches)"
timeout 400 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8,1 --poly3 --blocks 160 --opt mix_kernel=1 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_mfma.txt
echo "== sweep fused"
timeout 400 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8,1 --poly3 --blocks 160 --opt mix_kernel=2 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_fused.txt
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
for v in f_noepi f_noloads f_sameops; do
  echo "== variant $v"
  XL_LIBRARY_PATH=$V/lib$v.so timeout 300 python tools/group_sweep.py --clients 4096 --groups 8 --poly3 --blocks 160 --opt mix_kernel=2 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_$v.txt
done
echo "== PMC fused 4096 (launch-bound shape)"
EXTRA="--opt mix_kernel=2" bash tools/pmc_group.sh $TAG/pmc_fused4096 4096 8 optimized > $OUT/pmc_fused4096.log 2>&1
find $OUT/pmc_fused4096 -name "*.csv" -delete
python3 - $TAG <<'PY'
import json,sys
j=json.load(open("gpurun_out/%s/pmc_fused4096/pmc_group.json"%sys.argv[1]))
for k,d in j["per_dispatch_mean"].items():
    print(k, {c:d[c] for c in d if c in ("hbm_bytes","FETCH_SIZE","WRITE_SIZE","l2_hit_rate","TCC_HIT_sum","TCC_MISS_sum","TCC_REQ_sum","SQ_WAVE_CYCLES","SQ_WAIT_INST_ANY","wait_inst_over_wave_cycles","SQ_INSTS_VALU","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_INSTS_MFMA","SQ_VALU_MFMA_BUSY_CYCLES","SQ_BUSY_CU_CYCLES","SQ_WAVES")})
PY
