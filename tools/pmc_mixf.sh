#!/bin/bash
# tools/pmc_mixf.sh <tag> [clients] -- SQ counters of the float32 matrix-core mix launch on BASELINE config 5 (cf32 10 Msps, D = 100), 8 blocks per call
TAG=${1:-pmcf}; CLIENTS=${2:-1024}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/group_sweep.py --shape config5 --clients $CLIENTS --groups 8 --modes optimized --blocks 48"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in ("a", "b"):
    fs = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("xlp_mix"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        for n, v in c.items():
            v = v[2:] if len(v) > 4 else v
            res[k][n] = round(sum(v) / len(v), 1)
json.dump(res, open(out + "/pmc_mixf.json", "w"), indent=1)
for k, e in res.items():
    w = e.get("SQ_WAVES", 1); cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k, "waves", w, "kernel cycles %.0f K" % (cyc / 1e3))
    print("   per wave: VALU %.0f (of them MFMA %.0f), LDS %.0f, SALU %.0f" % (e.get("SQ_INSTS_VALU", 0) / w, e.get("SQ_INSTS_MFMA", 0) / w, e.get("SQ_INSTS_LDS", 0) / w, e.get("SQ_INSTS_SALU", 0) / w))
    if cyc:
        print("   matrix pipe busy %.0f %% of the kernel's cycles (per SIMD); waves resident per CU %.1f; waiting for an instruction %.0f %% of the wave cycles; LDS conflicts %.1f M of %.1f M cycles" % (
            100 * e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc, e.get("SQ_WAVE_CYCLES", 0) * 4 / cyc / 256, 100 * e.get("SQ_WAIT_INST_ANY", 0) / max(e.get("SQ_WAVE_CYCLES", 1), 1),
            e.get("SQ_LDS_BANK_CONFLICT", 0) / 1e6, e.get("SQ_LDS_IDX_ACTIVE", 0) / 1e6))
    print("   ", dict(e))
PY
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete 2>/dev/null
