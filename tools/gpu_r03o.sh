#!/bin/bash
# round 3, GPU session o: the final tree (size rule of the matrix-core classes lowered) -- tests, bench lines, kernel stats, configs, sweep
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03o; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" $OUT/pytest.log | head -5
( time timeout 600 python bench.py ) > $OUT/bench_n1.json 2> $OUT/bench_n1.err
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-spot --no-pmc > $OUT/kstats.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $OUT/kstats -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
rm -rf $OUT/kstats
timeout 600 python tools/measure_configs.py > $OUT/configs.json 2> $OUT/configs.err
timeout 300 python tools/group_sweep.py --clients 128,1024,4096 --groups 1,8,16 --poly3 --blocks 640 > $OUT/sweep_blocks_per_call.txt 2>&1
python - <<'PY'
import json
for f in ("bench_n1", "bench_driver_shape"):
    j=json.loads(open(f"gpurun_out/r03o/{f}.json").read().strip().splitlines()[-1])
    r=j["roofline"]
    print(f, j["value"], j["repeats"]["values"], "frac", r["frac"], "traffic", r["traffic"], r["traffic_source"][:30], "kernel_ms", r["kernel_ms"], "period", r["call_period_ms"])
    print("  ", {k:(v["ms"],v.get("hbm_bytes"),v.get("frac_hbm")) for k,v in r["per_kernel"].items()})
    print("  spot", j["parity_spot"]["ok"], j["parity_spot"]["clients"], j["parity_spot"]["max_rel"], "native", j["native"]["value"], j["native"]["parity_spot"]["bit_exact"] if j["native"]["parity_spot"] else None)
    for k,v in j["variants"].items(): print("   ", k[:70], v["value"], v["us_per_block"], v.get("roofline",{}).get("frac"))
PY
grep -v amdgpu $OUT/sweep_blocks_per_call.txt
