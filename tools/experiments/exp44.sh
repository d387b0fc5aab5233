#!/bin/bash
for e in 1 4 16 1000000; do echo "== timing every $e"; XL_TIMING_EVERY=$e python tools/sweep.py --clients 1024 --rates 5,1 --modes optimized,native --steps 200 2>&1 | grep -v amdgpu.ids | grep -v "^mode"; done
