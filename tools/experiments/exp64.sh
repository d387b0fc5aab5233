#!/bin/bash
# timeline of the mix launch at 1024 clients, M = 128 (1024 two-wave workgroups) and M = 256 (2048 one-wave workgroups)
OUT=gpurun_out/s64; mkdir -p $OUT
for M in 128 256; do
XL_EXP_POLY_M=$M XL_EXP_POLY_TRACE=$OUT/t$M.bin python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 3 2>&1 | grep optimized
python tools/poly_trace.py $OUT/t$M.bin 16 $((M*8)) | grep -v "^nco wave" | tee $OUT/trace_$M.txt
rm -f $OUT/t$M.bin
done
XL_EXP_POLY_M=128 XL_EXP_POLY_TRACE=$OUT/t.bin python tools/sweep.py --clients 4096 --rates 5 --modes optimized --steps 3 2>&1 | grep optimized
python tools/poly_trace.py $OUT/t.bin 64 4096 | grep -v "^nco wave" | tee $OUT/trace_4096_128.txt
