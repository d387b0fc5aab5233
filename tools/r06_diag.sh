#!/bin/bash
OUT=$1
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 300 python tools/r06_c5_diag.py 2>&1 | grep -v amdgpu.ids | grep "^n=" | cut -c1-30,100-170; done | tee $OUT/c5_diag.txt
