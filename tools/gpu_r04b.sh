#!/bin/bash
# round 4, session b: where does the fused launch spend its time?  Variants of xl_fused.hip (tools/experiments/build_variant.sh):
# no epilogue / no operand loads / every tile streaming the same (L2-resident) operands; counters of the full kernel.
TAG=r04b; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
for v in "" f_noepi f_noloads f_sameops f_noepi_sameops; do
  echo "== variant ${v:-full}"
  if [ -n "$v" ]; then export XL_LIBRARY_PATH=$V/lib$v.so; else unset XL_LIBRARY_PATH; fi
  timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 8 --poly3 --blocks 160 --opt mix_kernel=2 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_${v:-full}.txt
done
unset XL_LIBRARY_PATH
echo "== pytest fused"
timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=600 -k "fused or role_phases" > $OUT/pytest_fused.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_fused.txt; tail -5 $OUT/pytest_fused.txt
echo "== PMC fused 1024"
EXTRA="--opt mix_kernel=2" bash tools/pmc_group.sh $TAG/pmc_fused 1024 8 optimized > $OUT/pmc_fused.log 2>&1; tail -3 $OUT/pmc_fused.log
find $OUT/pmc_fused -name "*.csv" -delete
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r04b/pmc_fused/pmc_group.json"))
for k,d in j["per_dispatch_mean"].items():
    print(k, {c:d[c] for c in d if c in ("hbm_bytes","FETCH_SIZE","WRITE_SIZE","l2_hit_rate","TCC_HIT_sum","TCC_MISS_sum","TCC_REQ_sum","SQ_WAVE_CYCLES","SQ_WAIT_INST_ANY","wait_inst_over_wave_cycles","SQ_INSTS_VALU","SQ_LDS_BANK_CONFLICT","SQ_ACTIVE_INST_LDS","SQ_INSTS_MFMA","SQ_VALU_MFMA_BUSY_CYCLES","SQ_BUSY_CU_CYCLES","SQ_INSTS_VMEM_RD","SQ_IFETCH")})
PY
