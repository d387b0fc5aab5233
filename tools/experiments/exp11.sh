#!/bin/bash
OUT=gpurun_out/s11; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
echo "== sweep"; python tools/sweep.py --clients 64,512,1024,2048,4096 --rates 5,1 --modes optimized 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.log
echo "== native"; python tools/sweep.py --clients 1024 --rates 5,1 --modes native 2>&1 | grep -v amdgpu.ids | tail -2
echo "== bench"; timeout 600 python bench.py --steps 200 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
