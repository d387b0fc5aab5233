#!/bin/bash
# round 3, GPU session d: x86 modes, single-filter Q15 rework, latency test; configs table; python-stall evidence
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03d; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" $OUT/pytest.log | head -8
timeout 600 python tools/measure_configs.py > $OUT/configs.json 2> $OUT/configs.err; python -c "
import json;j=json.load(open('$OUT/configs.json'));print(json.dumps({k:j[k] for k in j if k.startswith('config1')},indent=0)[:1800])"
timeout 300 python tools/dropin_python_stall.py > $OUT/python_stall.json 2>/dev/null; cat $OUT/python_stall.json
