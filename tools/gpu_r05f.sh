#!/bin/bash
# round 5, session f: the bench line in the driver's shape (counter passes over three workloads per rocprofv3 process), rocprofv3
# kernel stats of the bench command and of BASELINE config 5.  Usage: gpurun --timeout 1500 -- 'bash tools/gpu_r05f.sh r05f'
TAG=${1:-r05f}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo) > $OUT/env.txt 2>&1
echo "== bench (driver shape)"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench.err ) 2>&1 | grep real
tail -3 $OUT/bench.err
python3 - $OUT <<'PY'
import json, sys
j = json.load(open(sys.argv[1] + "/bench_driver_shape.json"))
r = j["roofline"]
print(j["value"], j["ms_per_step"], j["config"]["us_per_block"], "frac", r["frac"], "useful", r.get("frac_algorithmic_shared"), "traffic", r["traffic"], r["traffic_source"][:60], "spot", j["parity_spot"]["ok"], "cpu", j["cpu_baseline"]["value"])
for k, v in r.get("per_kernel", {}).items():
    print("   ", k, v.get("ms"), v.get("hbm_bytes"), v.get("frac_hbm"))
for k, v in j["variants"].items():
    if "value" in v:
        print("  ", k[:90], v["value"], v["us_per_block"], (v.get("parity_spot") or {}).get("ok"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("frac_algorithmic_shared"))
print(json.dumps(j["variants"].get("config 5: cf32 10 Msps, D=100, 257 taps, 1024 clients", {}).get("roofline", {}))[:2500])
PY
echo "== rocprofv3 kernel stats of the bench command"
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-variants --no-spot --no-pmc > $OUT/prof_bench.json 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -8 $f; cp $f $OUT/rocprofv3_kernel_stats.csv; done
for f in $(find $OUT/prof -name "*kernel_trace.csv" | head -1); do python3 tools/timeline.py $f 20 3 > $OUT/timeline_bench_1024clients.txt; tail -3 $OUT/timeline_bench_1024clients.txt; done
rm -rf $OUT/prof
echo "== rocprofv3 kernel stats, config 5 at 1024 clients (8 blocks per call)"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof5 -o c5 -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --shape config5 --clients 1024 --groups 8 --modes optimized --blocks 640 > $OUT/prof5.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof5 -name "*kernel_stats*.csv" | head -1); do head -8 $f; cp $f $OUT/rocprofv3_kernel_stats_config5_1024clients.csv; done
rm -rf $OUT/prof5
grep optimized $OUT/prof5.log
