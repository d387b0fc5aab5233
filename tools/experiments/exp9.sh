#!/bin/bash
OUT=gpurun_out/s9; mkdir -p $OUT
for n in 1024 4096; do for h in 8 10; do
echo "== trace clients=$n H=$h"; XL_EXP_H=$h XL_EXP_TRACE=$OUT/trace_${n}_$h.bin python tools/sweep.py --clients $n --rates 5 --modes optimized --depths 1 --steps 3 2>&1 | grep -v amdgpu.ids | tail -1
python tools/trace_analyze.py $OUT/trace_${n}_$h.bin | tee $OUT/trace_${n}_$h.txt; rm -f $OUT/trace_${n}_$h.bin
done; done
