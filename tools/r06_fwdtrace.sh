#!/bin/bash
# forward launch: where its 12-17 us go.  (1) the kernel's own clock stamps per workgroup (-DXL_TUNING library, XL_EXP_POLY_TRACE_FWD);
# (2) anatomy: variant libraries with the sample loads / the transforms / the image stores compiled out (wrong results), alternating
OUT=$1; V=sdr-server_amd/build/variants
for shape in config5 server; do
  XL_LIBRARY_PATH=$V/libtune.so XL_EXP_POLY_TRACE=$OUT/fwd_trace_$shape.bin XL_EXP_POLY_TRACE_FWD=1 timeout 300 python tools/group_sweep.py --shape $shape --clients 1024 --groups 8 --modes optimized --blocks 64 2>&1 | grep optimized
  echo "# $shape, 1024 clients x 8 blocks"; python tools/fwd_trace.py $OUT/fwd_trace_$shape.bin 700
done | tee $OUT/forward_trace.txt
rm -f $OUT/fwd_trace_*.bin
for rep in 1 2; do
  for shape in config5 server; do
    timeout 200 python tools/group_sweep.py --shape $shape --clients 1024 --groups 8 --modes optimized --poly3 --blocks 1600 2>&1 | grep optimized | sed "s/^/shipped  $shape /"
    for v in NOLOAD NODFT NOSTORE NOLOAD_NODFT NOLOAD_NOSTORE ALL; do
      XL_LIBRARY_PATH=$V/libfwd_$v.so timeout 200 python tools/group_sweep.py --shape $shape --clients 1024 --groups 8 --modes optimized --poly3 --blocks 1600 2>&1 | grep optimized | sed "s/^/$v $shape /"
    done
  done
done | tee $OUT/forward_anatomy.txt
