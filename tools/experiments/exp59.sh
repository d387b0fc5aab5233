#!/bin/bash
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3
python tools/sweep.py --clients 1024 --rates 5,1 --modes optimized,native --steps 100 2>&1 | grep -v amdgpu.ids
XL_EXP_POLY=0 python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 100 2>&1 | grep -v amdgpu.ids | grep -v "^mode"
