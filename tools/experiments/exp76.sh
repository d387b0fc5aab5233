#!/bin/bash
# phases of the inverse launch (load / transforms / stores) at 1024 and 4096 clients
OUT=gpurun_out/s76; mkdir -p $OUT
for N in 1024 4096; do
XL_EXP_POLY_TRACE=$OUT/t.bin XL_EXP_POLY_TRACE_INV=1 python tools/sweep.py --clients $N --rates 5 --modes optimized --steps 3 2>&1 | grep optimized
python tools/inv_trace.py $OUT/t.bin $((27*N/32)) | tee $OUT/inv_$N.txt
done
rm -f $OUT/t.bin
