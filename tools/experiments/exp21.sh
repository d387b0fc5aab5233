#!/bin/bash
for pr in 3 2 1; do echo "== NCO role prio $pr"; XL_EXP_NCOPRIO=$pr python tools/sweep.py --clients 960,1024,2048 --rates 5,1 --modes optimized 2>&1 | grep -v amdgpu.ids | tail -6; done
