#!/bin/bash
# round 5, final evidence session: the whole GPU suite, smoke, the bench line in the driver's shape (counter passes inside the run),
# rocprofv3 kernel stats of the bench command and of BASELINE config 5 at 1024 clients, the kernel timeline of the headline shape.
# Usage: gpurun --timeout 1800 -- 'bash tools/gpu_r05_final.sh r05z'      (outputs: gpurun_out/<tag>/, copied into profiles/r05_*)
TAG=${1:-r05z}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo; /opt/rocm/bin/hipcc --version | head -2; python -c "import torch;print('torch', torch.__version__)") > $OUT/env.txt 2>&1
echo "== pytest -m gpu"
( time timeout 900 python -m pytest tests -m gpu -q --timeout=600 --durations=12 > $OUT/pytest_gpu.txt 2>&1 ) 2>&1 | grep real
echo "pytest exit ${PIPESTATUS[0]}" >> $OUT/pytest_gpu.txt
tail -4 $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $OUT/smoke.txt | tail -3
echo "== bench (driver shape)"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench.err ) 2>&1 | grep real
tail -3 $OUT/bench.err
python3 - $OUT <<'PY'
import json, sys
j = json.load(open(sys.argv[1] + "/bench_driver_shape.json"))
r = j["roofline"]
print(j["value"], j["ms_per_step"], j["config"]["us_per_block"], "frac", r["frac"], "useful", r.get("frac_algorithmic_shared"), "traffic", r["traffic"], r["traffic_source"][:60], "spot", j["parity_spot"]["ok"], "cpu", j["cpu_baseline"]["value"])
print("plan", j["config"].get("plan"))
for k, v in r.get("per_kernel", {}).items():
    print("   ", k, v.get("ms"), v.get("hbm_bytes"), v.get("frac_hbm"))
print("native", j["native"]["value"], j["native"]["us_per_block"], j["native"]["parity_spot"]["ok"])
for k, v in j["variants"].items():
    if "value" in v:
        print("  ", k[:90], v["value"], v.get("us_per_block"), (v.get("parity_spot") or {}).get("ok"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("frac_algorithmic_shared"))
    else:
        print("  ", k[:60], json.dumps(v)[:600])
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
echo "== rocprofv3 kernel stats, 4096 clients (8 blocks per call; the launch-bound shape)"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof4 -o c4 -- python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 4096 --groups 8 --modes optimized --blocks 320 > $OUT/prof4.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof4 -name "*kernel_stats*.csv" | head -1); do head -6 $f; cp $f $OUT/rocprofv3_kernel_stats_4096clients.csv; done
rm -rf $OUT/prof4
grep optimized $OUT/prof4.log
