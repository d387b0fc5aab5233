#!/bin/bash
# what the two-half mix launch's time is made of at BASELINE config 5 (13 k-blocks, 2 waves per SIMD): parts compiled out (wrong results)
OUT=$1; V=sdr-server_amd/build/variants
for rep in 1 2; do
  timeout 200 python tools/group_sweep.py --shape config5 --clients 1024,2048 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/full        /"
  for v in NOSTORE NOMFMA NOSTAGE NOOPERANDS; do
    XL_LIBRARY_PATH=$V/libmix_$v.so timeout 200 python tools/group_sweep.py --shape config5 --clients 1024,2048 --groups 8 --modes optimized --poly3 --blocks 320 2>&1 | grep optimized | sed "s/^/$(printf '%-12s' $v)/"
  done
done | tee $OUT/mix_anatomy_config5.txt
