#!/bin/bash
# round 3, GPU session b: register-transform inverse kernel -- parity suite, A/B timing vs the LDS kernel, LDS counters
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for inv in 0 1 0 1; do
  XL_EXP_INV=$inv timeout 300 python tools/group_sweep.py --clients 1024,4096 --groups 8 --poly3 --blocks 640 > $OUT/sweep_inv${inv}_$RANDOM.txt 2>&1
done
cat $OUT/sweep_inv*.txt | grep -v "^mode"
cd /tmp
for inv in 0 1; do
  XL_EXP_INV=$inv timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_lds_inv$inv -o p -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 8 --blocks 48 > $OUT/pmc_lds_inv$inv.log 2>&1
  XL_EXP_INV=$inv timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_inv$inv -o p -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 8 --blocks 48 > $OUT/pmc_fetch_inv$inv.log 2>&1
  XL_EXP_INV=$inv timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_inv$inv -o p -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 8 --blocks 48 > $OUT/pmc_write_inv$inv.log 2>&1
  XL_EXP_INV=$inv timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_inv$inv -o p -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups 8 --blocks 320 > $OUT/trace_inv$inv.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r03b"
for inv in (0, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("pmc_lds", "pmc_fetch", "pmc_write"):
        for f in glob.glob(f"{out}/{d}_inv{inv}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if k.startswith("xlp_") and "tables" not in k:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(inv, k, {c: round(sum(v[3:]) / max(len(v[3:]), 1), 1) for c, v in d.items()})
    for f in glob.glob(f"{out}/trace_inv{inv}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Name"].startswith(("xlp_", "void xlp_", "xl_nco", "void xl_")):
                print(inv, "stats", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
timeout 300 python tools/replan_cost.py > $OUT/replan.txt 2> $OUT/replan.err; cat $OUT/replan.txt; grep trial $OUT/replan.err | head -12
timeout 300 python tools/dropin_python_stall.py > $OUT/python_stall.json 2>/dev/null; cat $OUT/python_stall.json
