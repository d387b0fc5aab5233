#!/bin/bash
for SK in 0 256 264 512 1024; do echo "== FIR skip at $SK"; XL_EXP_FIRSKIP=$SK XL_EXP_POLY=0 python tools/sweep.py --clients 1024,2048 --rates 1,5 --modes optimized,native --steps 100 2>&1 | grep -v amdgpu.ids | grep -v "^mode"; done
