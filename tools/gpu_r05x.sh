#!/bin/bash
# round 5, session x: what the two-half mix launch's time is made of -- variant libraries with parts of xlp_mix_mfma_kernel compiled out
# (wrong results): no stores / no matrix instructions / no staging of the next pass / no operand loads, and pairs of them.
TAG=${1:-r05x}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
for rnd in 1 2; do
  timeout 200 python tools/group_sweep.py --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 160 2>&1 | grep optimized | sed "s/^/full              /"
  for v in NOSTORE NOMFMA NOSTAGE NOOPERANDS NOSTORE_NOSTAGE NOMFMA_NOSTAGE; do
    XL_TESTING=1 XL_LIBRARY_PATH=$V/libmix_$v.so timeout 200 python tools/group_sweep.py --clients 1024,4096 --groups 8 --modes optimized --poly3 --blocks 160 2>&1 | grep optimized | sed "s/^/$(printf '%-18s' $v)/"
  done
done | tee $OUT/mix_anatomy.txt
