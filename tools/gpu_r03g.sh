#!/bin/bash
# round 3, GPU session e: quad-lane register inverse kernel -- parity (polyphase tests), A/B timing and counters vs the LDS kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03g; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_batch_gpu.py -m gpu -x -q -k "group_of_blocks_polyphase" ) > $OUT/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" $OUT/pytest.log | head -8
for inv in 3 4 3 4 0; do
  XL_EXP_INV=$inv timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 8 --poly3 --blocks 640 > $OUT/sweep_inv${inv}_$RANDOM.txt 2>&1
done
cat $OUT/sweep_inv*.txt | grep -v "^mode" | grep -v amdgpu
cd /tmp
for inv in 4; do
  XL_EXP_INV=$inv timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_lds_inv$inv -o p -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 8 --blocks 48 > $OUT/pmc_lds_inv$inv.log 2>&1
  XL_EXP_INV=$inv timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_inv$inv -o p -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 8 --blocks 48 > $OUT/pmc_fetch_inv$inv.log 2>&1
  XL_EXP_INV=$inv timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_inv$inv -o p -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 8 --blocks 48 > $OUT/pmc_write_inv$inv.log 2>&1
  XL_EXP_INV=$inv timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_inv$inv -o p -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 8 --blocks 320 > $OUT/trace_inv$inv.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/r03g"
for inv in (4,):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("pmc_lds", "pmc_fetch", "pmc_write"):
        for f in glob.glob(f"{out}/{d}_inv{inv}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if k.startswith("xlp_inverse"):
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(inv, k, {c: round(sum(v[3:]) / max(len(v[3:]), 1), 1) for c, v in d.items()})
    for f in glob.glob(f"{out}/trace_inv{inv}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "xlp_" in r["Name"] or "xl_nco_chain" in r["Name"]:
                print(inv, "stats", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
