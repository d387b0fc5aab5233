#!/bin/bash
# what bounds a wave of the matrix-core mix kernel: rebuild with pieces removed (-DXLP_EXPT: 1 no X loads in the loop,
# 2 no R loads in the loop, 4 no Y stores; WRONG results), per-wave durations from the trace, M = 128
OUT=$GRAFT_REPO_ROOT/gpurun_out/s70; mkdir -p $OUT
export TMPDIR=/tmp
for E in 0 1 2 4 3 7; do
touch sdr-server_amd/csrc/xl_polyphase.hip
make -C sdr-server_amd/csrc EXTRA=-DXLP_EXPT=$E 2>&1 | grep -E "error" 
for N in 1024 256; do
XL_EXP_POLY_SLICES=12000,40000 python tools/sweep.py --clients $N --rates 5 --modes optimized --steps 100 2>&1 | grep optimized > $OUT/a.log
XL_EXP_POLY_SLICES=12000,40000 XL_EXP_POLY_TRACE=$OUT/t.bin python tools/sweep.py --clients $N --rates 5 --modes optimized --steps 3 > /dev/null 2>&1
echo "== expt $E clients $N: $(awk '{print $5, $10}' $OUT/a.log)  $(python tools/poly_trace.py $OUT/t.bin $((N/64)) $((N*2)) | grep durations)"
done; done
python tools/poly_place.py $OUT/t.bin 512
rm -f $OUT/t.bin
