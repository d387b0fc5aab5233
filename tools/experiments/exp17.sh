#!/bin/bash
OUT=gpurun_out/s19; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -5 | tee $OUT/pytest.log
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== sweep"; python tools/sweep.py --clients 1024 --rates 5,1 --modes optimized,native 2>&1 | grep -v amdgpu.ids
