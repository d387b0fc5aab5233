#!/bin/bash
OUT=gpurun_out/s41; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
echo "== bench"; timeout 600 python bench.py --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
bash tools/pmc_traffic.sh s41 | tail -60
