#!/bin/bash
# tools/pmc_traffic.sh <tag> -- HBM traffic per block of the bench workload (1024 clients, 505 taps, optimized) from the
# PMC counters FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, MI355X_MICROARCH.md HBM section), for the
# polyphase path (three launches) and for the direct FIR kernel (XL_EXP_POLY=0).  Writes <out>/pmc_traffic.json.
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for path in poly direct; do
  for c in FETCH_SIZE WRITE_SIZE; do
    if [ $path = direct ]; then export XL_EXP_POLY=0; else unset XL_EXP_POLY; fi
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${path}_$c -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 8 > $OUT/${path}_$c.log 2>&1
  done
done
unset XL_EXP_POLY
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for path in ("poly", "direct"):
    per = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"{out}/{path}_{c}/**/*counter_collection.csv", recursive=True)
        if not fs: continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] != c: continue
            k = r["Kernel_Name"].split("(")[0]
            k = k[5:] if k.startswith("void ") else k
            if k.startswith("xlp_") and "tables" not in k or k.startswith("xl_fir_kernel"):
                agg[k].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            v = v[2:] if len(v) > 4 else v          # skip the first blocks (stand-alone NCO tabulation, cold caches)
            per[k][c + "_KiB"] = round(sum(v) / len(v), 1)
    tot = 0
    for k, d in per.items():
        d["hbm_bytes"] = int((2 * d.get("FETCH_SIZE_KiB", 0) + d.get("WRITE_SIZE_KiB", 0)) * 1024)
        tot += d["hbm_bytes"]
    res[path] = {"kernels": per, "hbm_bytes_per_block": tot}
res["correction"] = "gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024"
res["command"] = "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 8   (XL_EXP_POLY=0 for the direct FIR kernel)"
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
# the digest bench.py reads (copy to profiles/pmc_latest.json)
latest = {"source": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over python tools/sweep.py --clients 1024 --rates 5 --modes optimized",
          "correction": res["correction"],
          "hbm_bytes_per_block_polyphase": res.get("poly", {}).get("hbm_bytes_per_block"),
          "polyphase_kernels": res.get("poly", {}).get("kernels"),
          "hbm_bytes_per_launch": res.get("direct", {}).get("hbm_bytes_per_block"),
          "direct_kernel": res.get("direct", {}).get("kernels")}
json.dump(latest, open(f"{out}/pmc_latest.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
