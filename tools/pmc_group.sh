#!/bin/bash
# tools/pmc_group.sh <tag> [clients] [group] [mode] -- PMC counters of the bench workload's launches (1024 clients, 505 taps,
# 8 blocks per call): HBM traffic from FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, MI355X_MICROARCH.md HBM
# section: bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB on gfx950) and the SQ / TCC counters of the three polyphase kernels
# (or the direct FIR kernel in native mode).  Writes <out>/pmc_group.json (+ pmc_latest.json, the digest bench.py reads).
TAG=${1:-pmcg}; CLIENTS=${2:-1024}; G=${3:-8}; MODE=${4:-optimized}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients $CLIENTS --groups $G --modes $MODE --blocks 48 $EXTRA"  # EXTRA (env): more group_sweep options, e.g. "--opt mix_kernel=2"
run() { n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o p -- $CMD > $OUT/$n.log 2>&1
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
run sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_IFETCH
run sq4 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
python3 - $OUT $CLIENTS $G $MODE "$CMD" <<'PY'
import csv, glob, json, sys, collections
out, clients, G, mode, cmd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
per = collections.defaultdict(dict)
for n in ("fetch", "write", "sq1", "sq2", "sq3", "sq4", "tcc"):
    fs = glob.glob(f"{out}/{n}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        k = k[5:] if k.startswith("void ") else k
        if (k.startswith("xlp_") and "tables" not in k) or k.startswith("xl_fir_kernel"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            v = v[3:] if len(v) > 6 else v  # skip the first calls (stand-alone NCO tabulation, plan rebuild, cold caches)
            per[k][c] = round(sum(v) / len(v), 1)
            per[k]["dispatches_averaged"] = len(v)
tot = 0
for k, d in per.items():
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        d["hbm_bytes"] = int((2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024)
        tot += d["hbm_bytes"]
    if d.get("SQ_BUSY_CYCLES") and d.get("SQ_ACTIVE_INST_VALU"):
        # SQ_ACTIVE_INST_* count quad-cycles summed over the SIMDs; SQ_BUSY_CYCLES per SE/XCD aggregate -- report the ratios the guide uses
        d["valu_active_over_wave_cycles"] = round(d["SQ_ACTIVE_INST_VALU"] / max(d.get("SQ_WAVE_CYCLES", 1), 1), 4)
        d["wait_inst_over_wave_cycles"] = round(d.get("SQ_WAIT_INST_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1), 4)
    if d.get("TCC_HIT_sum") is not None and d.get("TCC_MISS_sum") is not None:
        d["l2_hit_rate"] = round(d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1), 4)
res = {"workload": f"{clients} clients x 48 kHz, 505 taps, {mode}, {G} blocks per call", "command": "rocprofv3 --pmc <set> --kernel-trace -- " + cmd,
       "counter_sets": "one rocprofv3 pass per set: FETCH_SIZE | WRITE_SIZE | SQ set 1 | SQ set 2 | SQ set 3 | TCC",
       "correction": "gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
       "per_dispatch_mean": per, "hbm_bytes_per_call": tot, "hbm_bytes_per_block": int(tot / G)}
json.dump(res, open(f"{out}/pmc_group.json", "w"), indent=1)
if clients == 1024 and G == 8 and not __import__('os').environ.get('EXTRA'):
    key = "hbm_bytes_per_call_polyphase" if mode == "optimized" else "hbm_bytes_per_call_direct"
    latest = {}
    try:
        latest = json.load(open(f"{out}/../../profiles/pmc_latest.json"))
    except Exception:
        pass
    latest.update({"source": f"tools/pmc_group.sh (session {out.split('/')[-1]}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over python tools/group_sweep.py --clients 1024 --groups 8",
                   "correction": res["correction"], key: tot, key.replace("per_call", "kernels"): {k: d.get("hbm_bytes") for k, d in per.items()}})
    json.dump(latest, open(f"{out}/pmc_latest.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
