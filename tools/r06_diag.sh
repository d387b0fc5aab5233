#!/bin/bash
OUT=$1
timeout 900 python tools/inverse_midband.py --rounds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/inverse_midband.txt
