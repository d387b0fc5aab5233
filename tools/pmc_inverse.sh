#!/bin/bash
# tools/pmc_inverse.sh <tag> [clients] -- SQ counters of the inverse launch's kernels (option inverse_kernel = 3, 5, 7) on the bench shape
# at <clients> clients, 8 blocks per call: instructions issued per launch by class, busy / wait cycles, LDS bank conflicts.
TAG=${1:-pmci}; CLIENTS=${2:-4096}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for inv in ${INVS:-3 5 7}; do
  CMD="python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients $CLIENTS --groups 8 --modes optimized --blocks 48"
  XL_EXP_INV=$inv timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a$inv -o p -- $CMD > $OUT/a$inv.log 2>&1
  XL_EXP_INV=$inv timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/b$inv -o p -- $CMD > $OUT/b$inv.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(out + "/[ab]*/")):
    inv = d.rstrip("/").split("/")[-1][1:]
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("xlp_inverse"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        e = res.setdefault(f"inverse_kernel={inv}: {k}", {})
        for n, v in c.items():
            v = v[2:] if len(v) > 4 else v
            e[n] = round(sum(v) / len(v), 1)
        e["dispatches_averaged"] = len(v)
json.dump(res, open(out + "/pmc_inverse.json", "w"), indent=1)
for k, e in res.items():
    w = e.get("SQ_WAVES", 1)
    print(k)
    print("   per wave: VALU %.0f  LDS %.0f  SALU %.0f  VMEM rd %.0f wr %.0f" % tuple(e.get(n, 0) / w for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")), " waves", w)
    print("   ", {n: v for n, v in e.items() if n not in ("SQ_WAVES",)})
PY
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete 2>/dev/null
