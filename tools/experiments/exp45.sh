#!/bin/bash
OUT=gpurun_out/s45; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
echo "== bench"; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline'].get('kernels_ms')); [print(k, v['value'], v['ms_per_step']) for k,v in d['variants'].items()]"; tail -2 $OUT/bench.err
