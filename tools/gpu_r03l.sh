#!/bin/bash
# round 3, GPU session l: matrix-core mix as a software pipeline (rows of the next pass staged before this pass's stores)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03l; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_batch_gpu.py -m gpu -x -q -k "polyphase or group_of_blocks or bench_shape or matrix_core or staggered" ) > $OUT/pytest_poly.log 2>&1
tail -4 $OUT/pytest_poly.log
for pp in 8 16; do
  XL_EXP_MIX=1 XL_EXP_MIX_PP=$pp timeout 300 python tools/group_sweep.py --clients 1024,2048,4096 --groups 8 --poly3 --blocks 640 > $OUT/sweep_mix1_pp$pp.txt 2>&1
  echo "== mix 1 pp $pp"; grep -v "amdgpu" $OUT/sweep_mix1_pp$pp.txt
done
timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-variants > $OUT/bench_quick.json 2> $OUT/bench_quick.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r03l/bench_quick.json").read().strip().splitlines()[-1])
print("bench", j["value"], j["repeats"]["values"], j["roofline"]["kernel_ms"], j["roofline"]["call_period_ms"], {k:v["ms"] for k,v in j["roofline"]["per_kernel"].items()})
PY
