#!/bin/bash
# round 3, GPU session p: the final tree -- tests, smoke, bench lines, kernel stats, counters of the polyphase launches
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03p; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" $OUT/pytest.log | head -5
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log | grep -v amdgpu
( time timeout 600 python bench.py ) > $OUT/bench_n1.json 2> $OUT/bench_n1.err
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-spot --no-pmc > $OUT/kstats.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $OUT/kstats -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
rm -rf $OUT/kstats
bash tools/pmc_group.sh r03p_pmc_opt 1024 8 optimized > $OUT/pmc_opt.log 2>&1
cp gpurun_out/r03p_pmc_opt/pmc_group.json $OUT/pmc_polyphase_1024clients_8blocks.json 2>/dev/null
rm -rf gpurun_out/r03p_pmc_opt/*/
python - <<'PY'
import json
for f in ("bench_n1", "bench_driver_shape"):
    j=json.loads(open(f"gpurun_out/r03p/{f}.json").read().strip().splitlines()[-1])
    r=j["roofline"]
    print(f, j["value"], j["repeats"]["values"], j["scaling"], "frac", r["frac"], "traffic", r["traffic"], r["traffic_source"][:30], "kernel_ms", r["kernel_ms"], "period", r["call_period_ms"])
    print("  ", {k:(v["ms"],v.get("hbm_bytes"),v.get("frac_hbm")) for k,v in r["per_kernel"].items()})
    print("  spot", j["parity_spot"]["ok"], j["parity_spot"]["clients"], j["parity_spot"]["max_rel"], "native", j["native"]["value"], j["native"]["parity_spot"]["bit_exact"] if j["native"]["parity_spot"] else None)
    for k,v in j["variants"].items(): print("   ", k[:70], v["value"], v["us_per_block"], v.get("roofline",{}).get("frac"))
j=json.load(open("gpurun_out/r03p/pmc_polyphase_1024clients_8blocks.json"))
for k,v in j["per_dispatch_mean"].items(): print(k, {c:v.get(c) for c in ("SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_INSTS_VALU","SQ_INSTS_MFMA","hbm_bytes")})
PY
