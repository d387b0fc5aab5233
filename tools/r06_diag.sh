#!/bin/bash
OUT=$1; V=sdr-server_amd/build/variants
for v in KM KMSCALAR; do
  export XL_LIBRARY_PATH=$V/lib$v.so
  for rep in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python tools/r06_c5_diag.py 2>&1 | grep -v amdgpu.ids | grep "^n=" | sed "s/^/$v: /" | cut -c1-14,100-170; done
done | tee $OUT/c5_diag.txt
