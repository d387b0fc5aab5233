#!/bin/bash
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -2
echo "== native"; python tools/sweep.py --clients 64,1024,4096 --rates 5,1 --modes native 2>&1 | grep -v amdgpu.ids
