#!/bin/bash
OUT=$1; V=sdr-server_amd/build/variants
for shape in config5 server; do
  XL_LIBRARY_PATH=$V/libtune.so XL_EXP_POLY_TRACE=$OUT/mix_trace.bin timeout 300 python tools/group_sweep.py --shape $shape --clients 1024 --groups 8 --modes optimized --blocks 64 2>&1 | grep optimized
  echo "# $shape 1024 clients x 8 blocks: the mix launch with the forward role"; python tools/fwdrole_trace.py $OUT/mix_trace.bin
  XL_EXP_FWD_IN_MIX=0 XL_LIBRARY_PATH=$V/libtune.so XL_EXP_POLY_TRACE=$OUT/mix_trace.bin timeout 300 python tools/group_sweep.py --shape $shape --clients 1024 --groups 8 --modes optimized --blocks 64 2>&1 | grep optimized
  echo "# $shape 1024 clients x 8 blocks: the mix launch alone"; python tools/fwdrole_trace.py $OUT/mix_trace.bin
done | tee $OUT/mix_trace_fwdrole.txt
rm -f $OUT/mix_trace.bin
