#!/bin/bash
# where the two waves of a mix workgroup land (M = 128, 1024 clients): per-wave placement from the trace
OUT=gpurun_out/s66; mkdir -p $OUT
XL_EXP_POLY_M=128 XL_EXP_POLY_TRACE=$OUT/t.bin python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 3 2>&1 | grep optimized
python tools/poly_place.py $OUT/t.bin 2048 | tee $OUT/place_128.txt
XL_EXP_POLY_M=256 XL_EXP_POLY_TRACE=$OUT/t.bin python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 3 2>&1 | grep optimized
python tools/poly_place.py $OUT/t.bin 2048 | tee $OUT/place_256.txt
rm -f $OUT/t.bin
