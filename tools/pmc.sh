#!/bin/bash
# tools/pmc.sh <tag> <env-assignments...> -- PMC passes over the FIR kernel (4096- and 1024-client runs)
TAG=${1:-pmc}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|GRBM|TA|TD)_[A-Z0-9_]+" | sort -u > $OUT/counters.txt
wc -l $OUT/counters.txt
run() { # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients ${CLIENTS:-1024} --rates ${RATES:-5} --steps 6 > $OUT/$n.log 2>&1
  f=$(find $OUT/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")[:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "fir" not in k and "nco" not in k and "xlp" not in k: continue
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
run sq3 SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_LDS_UNALIGNED_STALL
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
