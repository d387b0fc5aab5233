#!/bin/bash
# round 3, GPU session m: the final build (matrix-core mix, pipelined; one-block calls on the side stream)'s evidence -- test log, bench lines, kernel stats, counters, configs, sweeps
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
{ rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8; lscpu | grep -E "Model name|^CPU\(s\)|Socket|Core"; python -c "import torch;print('torch',torch.__version__)"; hipcc --version | head -2; } > $OUT/env.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" $OUT/pytest.log | head -5
( time timeout 600 python bench.py ) > $OUT/bench_n1.json 2> $OUT/bench_n1.err
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-spot --no-pmc > $OUT/kstats.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $OUT/kstats -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
bash tools/pmc_group.sh r03m_pmc_opt 1024 8 optimized > $OUT/pmc_opt.log 2>&1
bash tools/pmc_group.sh r03m_pmc_nat 1024 8 native > $OUT/pmc_nat.log 2>&1
cp gpurun_out/r03m_pmc_opt/pmc_group.json $OUT/pmc_polyphase_1024clients_8blocks.json 2>/dev/null
cp gpurun_out/r03m_pmc_nat/pmc_group.json $OUT/pmc_native_direct_1024clients_8blocks.json 2>/dev/null
cp gpurun_out/r03m_pmc_nat/pmc_latest.json $OUT/pmc_latest.json 2>/dev/null || cp gpurun_out/r03m_pmc_opt/pmc_latest.json $OUT/pmc_latest.json 2>/dev/null
timeout 600 python tools/measure_configs.py > $OUT/configs.json 2> $OUT/configs.err
timeout 300 python tools/group_sweep.py --clients 128,1024,4096 --groups 1,8,16 --poly3 --blocks 640 > $OUT/sweep_blocks_per_call.txt 2>&1
timeout 200 python tools/replan_cost.py > $OUT/replan.txt 2> /dev/null
python - <<'PY'
import json
for f in ("bench_n1", "bench_driver_shape"):
    j=json.loads(open(f"gpurun_out/r03m/{f}.json").read().strip().splitlines()[-1])
    r=j["roofline"]
    print(f, j["value"], j["repeats"]["values"], "frac", r["frac"], "traffic", r["traffic"], r["traffic_source"][:30], "kernel_ms", r["kernel_ms"], "period", r["call_period_ms"])
    print("  ", {k:(v["ms"],v.get("hbm_bytes"),v.get("frac_hbm"),v.get("frac_fp32")) for k,v in r["per_kernel"].items()})
    print("  spot", j["parity_spot"]["ok"], j["parity_spot"]["clients"], j["parity_spot"]["max_rel"], "native", j["native"]["value"], j["native"]["parity_spot"]["bit_exact"] if j["native"]["parity_spot"] else None)
    for k,v in j["variants"].items(): print("   ", k[:70], v["value"], v["us_per_block"])
    print("  cpu", j.get("cpu_baseline",{}).get("value"), j.get("cpu_baseline",{}).get("single_thread_value"))
PY
cat $OUT/sweep_blocks_per_call.txt | grep -v amdgpu; cat $OUT/replan.txt
