#!/bin/bash
# round 4, session l: 48-bit Y in the pair layout: parity, A/B against float32 pairs on one box, counters; one block per call at 4096.
TAG=${1:-r04l}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=400 -k "polyphase or group_of_blocks or bench_shape or 2048_clients or 4096_clients" > $OUT/pytest_poly.txt 2>&1
rc=$?; tail -4 $OUT/pytest_poly.txt; if [ $rc -ne 0 ]; then tail -40 $OUT/pytest_poly.txt; exit 1; fi
for y in 0 1 0 1; do
  echo "== y_format=$y"
  timeout 300 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --poly3 --blocks 240 --opt y_format=$y 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep_y$y.txt
done
timeout 200 python tools/group_sweep.py --clients 4096 --groups 1 --poly3 --blocks 160 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_4096_g1.txt
cd /tmp
for y in 1; do
  CMD="python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 4096 --groups 8 --modes optimized --blocks 48 --opt y_format=$y"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_y${y}_$c -o p -- $CMD > $OUT/pmc_y${y}_$c.log 2>&1
  done
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for y in (1,):
    per = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{out}/pmc_y{y}_{c}/**/*counter_collection.csv", recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if k.startswith("xlp_") and "tables" not in k and r["Counter_Name"] == c:
                    agg[k].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                v = v[3:] if len(v) > 6 else v
                per[k][c] = sum(v) / len(v)
    tot = 0
    for k, d in per.items():
        b = (2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024
        tot += b
        print(f"y_format={y} {k:45s} fetch {d.get('FETCH_SIZE',0):10.0f} KiB write {d.get('WRITE_SIZE',0):10.0f} KiB  -> {b/1e6:8.1f} MB")
    print(f"y_format={y} total per call (4096 clients, 8 blocks) {tot/1e6:.1f} MB = {tot/4e6:.1f} MB per 1024 clients")
PY
find $OUT -name "*.csv" -delete
