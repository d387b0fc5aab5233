#!/bin/bash
# round 3, GPU session a: test suite, the reworked bench line (in-run PMC), drop-in latency distribution, re-plan cost breakdown
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03a; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
( time timeout 600 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
L=sdr-server_amd/build/dropin_latency
for v in optimized native; do
  XL_DROPIN_TRACE_US=500 timeout 120 $L $v 10000 > $OUT/lat_${v}_default.json 2> $OUT/lat_${v}_default.err
done
XL_DROPIN_TRACE_US=500 XL_EXP_NOLOOKAHEAD=1 timeout 120 $L optimized 10000 > $OUT/lat_optimized_nolookahead.json 2> $OUT/lat_optimized_nolookahead.err
XL_DROPIN_TRACE_US=500 XL_EXP_DROPIN_COPY=1 timeout 120 $L optimized 10000 > $OUT/lat_optimized_copy.json 2> $OUT/lat_optimized_copy.err
XL_DROPIN_TRACE_US=500 timeout 120 $L optimized 3000 2000 > $OUT/lat_optimized_paced2ms.json 2> $OUT/lat_optimized_paced2ms.err
XL_DROPIN_TRACE_US=500 AMD_LOG_LEVEL=0 HIP_FORCE_DEV_KERNARG=1 timeout 120 $L optimized 10000 > $OUT/lat_optimized_devkernarg.json 2> $OUT/lat_optimized_devkernarg.err
cat $OUT/lat_*.json
grep -h "slow call" $OUT/lat_*.err | head -40
XL_EXP_PLAN_TIMING=1 timeout 300 python tools/replan_cost.py > $OUT/replan.txt 2> $OUT/replan_plan_timing.txt
cat $OUT/replan.txt
