#!/bin/bash
# round 5, session e: the slimmed library (two mix kernels on the matrix cores, two inverse kernels, 16-segment passes): whole GPU suite,
# smoke, the bench line in the driver's shape (new: inverse A/B, config 5 with its own roofline, host-delivered outputs, all-float32).
# Usage: gpurun --timeout 1800 -- 'bash tools/gpu_r05e.sh r05e'
TAG=${1:-r05e}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo) > $OUT/env.txt 2>&1
echo "== pytest -m gpu"
( time timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 ) > $OUT/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.txt
grep -E "passed|failed|exit|real|Error" $OUT/pytest_gpu.txt | tail -6
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $OUT/smoke.txt
echo "== bench (driver shape)"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench.err ) 2>&1 | grep real
tail -5 $OUT/bench.err
python3 - $OUT <<'PY'
import json, sys
j = json.load(open(sys.argv[1] + "/bench_driver_shape.json"))
r = j["roofline"]
print(j["value"], j["ms_per_step"], j["config"]["us_per_block"], j["dtype"], "frac", r["frac"], "useful", r.get("frac_algorithmic_shared"), "traffic", r["traffic"], "spot", j["parity_spot"]["ok"], j["parity_spot"]["max_rel"], "cpu", j["cpu_baseline"]["value"])
print("native", j["native"]["value"], j["native"]["us_per_block"], j["native"]["parity_spot"]["ok"])
for k, v in j["variants"].items():
    if "value" in v:
        print("  ", k[:90], v["value"], v["us_per_block"], (v.get("parity_spot") or {}).get("ok"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("frac_algorithmic_shared"))
    else:
        print("  ", k[:90], json.dumps(v)[:700])
print(json.dumps(j["all_f32"])[:700])
print(json.dumps(j["variants"].get("config 5: cf32 10 Msps, D=100, 257 taps, 1024 clients", {}).get("roofline", {}))[:1500])
PY
