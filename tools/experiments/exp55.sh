#!/bin/bash
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -4
echo "== single filter, look-ahead on"; python tools/scratch/sf.py 2>&1 | grep -v amdgpu
echo "== single filter, look-ahead off"; XL_EXP_NOLOOKAHEAD=1 python tools/scratch/sf.py 2>&1 | grep -v amdgpu
