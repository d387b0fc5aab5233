#!/bin/bash
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -2
echo "== sweep"; python tools/sweep.py --clients 64,1024,4096 --rates 5,1 --modes optimized,native 2>&1 | grep -v amdgpu.ids
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
