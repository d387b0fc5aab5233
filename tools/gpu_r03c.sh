#!/bin/bash
# round 3, GPU session c: incremental re-plan -- parity suite (incl. the churn test), re-plan cost, where the 20 ms after a
# join at 1024 clients went (kernel trace of one join), bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
XL_EXP_PLAN_TIMING=1 timeout 300 python tools/replan_cost.py > $OUT/replan.txt 2> $OUT/replan.err; cat $OUT/replan.txt; grep trial $OUT/replan.err | head -12
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r03c/plan_breakdown.txt
import re
cur=[];blocks=[]
for l in open("gpurun_out/r03c/replan.err"):
    m=re.match(r"plan: (.*?)\s+([\d.]+) ms",l)
    if m:
        cur.append((m.group(1).strip(),float(m.group(2))))
        if m.group(1).strip()=="trim": blocks.append(cur); cur=[]
for i,b in enumerate(blocks):
    print(i, round(sum(v for _,v in b),2), " ".join(f"{k.split()[0]}={v:.2f}" for k,v in b))
PY
cat $OUT/plan_breakdown.txt | tail -24
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_replan -o p -- python $GRAFT_REPO_ROOT/tools/replan_cost.py > $OUT/trace_replan.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r03c/trace_replan/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)
    for r in rows[:12]:
        print(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3), "ms", r["Kernel_Name"][:70], "grid", r.get("Grid_Size"), "wg", r.get("Workgroup_Size"))
PY
( time timeout 600 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r03c/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["repeats"]["values"], j["roofline"]["frac"], j["roofline"]["traffic"], j["roofline"]["kernel_ms"], j["roofline"]["call_period_ms"])
print({k:(v["ms"],v.get("frac_hbm"),v.get("frac_fp32")) for k,v in j["roofline"]["per_kernel"].items()})
print(j["parity_spot"]["ok"], j["parity_spot"]["max_rel"], j["parity_spot"]["seconds"])
PY
