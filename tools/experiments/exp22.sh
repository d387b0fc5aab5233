#!/bin/bash
for rep in 1 2; do
for v in "" "XL_EXP_PRIO4=1" "XL_EXP_PRIOQUARTERS=1" "XL_EXP_FLATPRIO=1"; do echo "== [$v]"; env $v python tools/sweep.py --clients 960,1024,2048 --rates 5 --modes optimized --steps 80 2>&1 | grep -v amdgpu.ids | tail -3; done
done
