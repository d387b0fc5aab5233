#!/bin/bash
# matrix-core mix launch at 1024 clients: timeline with and without the skipped slot, chain waves included
OUT=gpurun_out/s71; mkdir -p $OUT
touch sdr-server_amd/csrc/xl_polyphase.hip; make -C sdr-server_amd/csrc 2>&1 | grep -E "error"
for V in "XL_EXP_POLY_EXP=0" "XL_EXP_POLY_EXP=16" "XL_EXP_MIXSKIP=2048"; do
echo "=== $V"
env $V XL_EXP_POLY_SLICES=12000,40000 XL_EXP_POLY_TRACE=$OUT/t.bin python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 3 2>&1 | grep optimized
python tools/poly_trace.py $OUT/t.bin 16 2048 | cut -c1-250 | head -24
python tools/poly_place.py $OUT/t.bin 2048 | head -6
done
rm -f $OUT/t.bin
