#!/bin/bash
# tools/mall_calibration.sh <tag> -- what do FETCH_SIZE / WRITE_SIZE count?    The mix -> inverse pair at call sizes
# whose mixed-spectra image Y is 28 / 113 / 226 / 452 MB (1 / 4 / 8 / 16 blocks per call, 1024 clients) -- below and above the 256 MiB
# Infinity Cache: kernel durations per block next to the fabric-side counters (FETCH_SIZE, WRITE_SIZE) and whatever DRAM-side /
# MALL counters this rocprofv3 exposes; the same with the Y stores temporal instead of non-temporal (a variant library).
TAG=${1:-mall}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -iE "DRAM|MALL|EA0_RDREQ|EA0_WRREQ|TCC_EA|HBM|IOMMU|_32B|_64B|PROBE" | head -60 > $OUT/counters_avail.txt
wc -l $OUT/counters_avail.txt
DRAMSET=$(grep -oE "TCC_EA0_(RD|WR)REQ_DRAM_sum|TCC_EA0_RDREQ_DRAM|TCC_EA0_WRREQ_DRAM" $OUT/counters_avail.txt | sort -u | head -4 | tr '\n' ' ')
echo "DRAM-side counters found: [$DRAMSET]"
run() { n=$1; lib=$2; g=$3; shift 3
  XL_TESTING=1 XL_LIBRARY_PATH=$lib timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o p -- \
    python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 1024 --groups $g --modes optimized --blocks $((g*12)) > $OUT/$n.log 2>&1
}
LIB=$GRAFT_REPO_ROOT/sdr-server_amd/lib/libxlating_hip.so
LIBT=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants/libp_ytemporal.so
for g in 1 4 8 16; do
  run nt_g${g}_fetch $LIB $g FETCH_SIZE
  run nt_g${g}_write $LIB $g WRITE_SIZE
  run nt_g${g}_tcc $LIB $g TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum
  [ -n "$DRAMSET" ] && run nt_g${g}_dram $LIB $g $DRAMSET
done
for g in 4 16; do
  run t_g${g}_fetch $LIBT $g FETCH_SIZE
  run t_g${g}_write $LIBT $g WRITE_SIZE
  [ -n "$DRAMSET" ] && run t_g${g}_dram $LIBT $g $DRAMSET
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(dict))  # run -> kernel -> counter
for d in sorted(glob.glob(f"{out}/*_g*_*")):
    if not os.path.isdir(d): continue
    name = os.path.basename(d); run = name.rsplit("_", 1)[0]
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if k in ("xlp_mix_mfma_kernel", "xlp_inverse_kernel", "xlp_forward_kernel"):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, dd in agg.items():
            for c, v in dd.items():
                v = v[3:] if len(v) > 6 else v
                res[run][k][c] = sum(v) / len(v)
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[:1]:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            v = v[3:] if len(v) > 6 else v
            if k in res[run] or k.startswith("xlp_"): res[run][k].setdefault("us", []).append(sum(v) / len(v))
for run in sorted(res):
    g = int(run.split("_g")[1])
    for k in ("xlp_mix_mfma_kernel", "xlp_inverse_kernel"):
        d = res[run].get(k, {})
        us = d.get("us", [0]); us = sum(us) / max(len(us), 1)
        line = f"{run:8s} {k:22s} {us / g:7.2f} us/block"
        for c in sorted(d):
            if c != "us": line += f" | {c} {d[c]:.0f}"
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            line += f" | fabric MB/block {(2 * d.get('FETCH_SIZE', 0) + d.get('WRITE_SIZE', 0)) * 1024 / 1e6 / g:.1f}"
        print(line)
PY
find $OUT -name "*.csv" -delete
