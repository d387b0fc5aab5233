#!/bin/bash
# round 4, final evidence session: the whole GPU suite, smoke, the bench line in the driver's shape, rocprofv3 kernel stats of the bench
# command, per-config numbers.  Usage: gpurun --timeout 1500 -- 'bash tools/gpu_r04_final.sh r04z'
TAG=${1:-r04z}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo) > $OUT/env.txt 2>&1
echo "== pytest -m gpu"
( time timeout 1000 python -m pytest tests -m gpu -q -x --timeout=600 ) > $OUT/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.txt
grep -E "passed|failed|exit|real" $OUT/pytest_gpu.txt | tail -4
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $OUT/smoke.txt
echo "== bench (driver shape)"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench.err ) 2>&1 | grep real
python3 - $OUT <<'PY'
import json, sys
j = json.load(open(sys.argv[1] + "/bench_driver_shape.json"))
print(j["value"], j["ms_per_step"], j["config"]["us_per_block"], j["dtype"], "frac", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], "spot", j["parity_spot"]["ok"], j["parity_spot"]["max_rel"], "cpu", j["cpu_baseline"]["value"])
print("native", j["native"]["value"], j["native"]["us_per_block"], j["native"]["parity_spot"]["ok"])
for k, v in j["variants"].items():
    print("  ", k[:90], v["value"], v["us_per_block"], (v.get("parity_spot") or {}).get("ok"), (v.get("roofline") or {}).get("frac"))
print(json.dumps(j["all_f32"])[:600])
PY
echo "== rocprofv3 kernel stats of the bench command"
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-variants --no-spot --no-pmc > $OUT/prof_bench.json 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -8 $f; cp $f $OUT/rocprofv3_kernel_stats.csv; done
for f in $(find $OUT/prof -name "*kernel_trace.csv" | head -1); do python3 tools/timeline.py $f 20 3 > $OUT/timeline_bench_1024clients.txt; tail -3 $OUT/timeline_bench_1024clients.txt; done
rm -rf $OUT/prof
echo "== configs"
timeout 500 python tools/measure_configs.py 2>/dev/null > $OUT/configs.json; head -c 1200 $OUT/configs.json; echo
