#!/bin/bash
# small classes (128 / 256 / 512 clients): the launches are bound by their NCO slices; which split is best there?
OUT=$GRAFT_REPO_ROOT/gpurun_out/s79; mkdir -p $OUT
for N in 128 256 512; do
for SL in "8000,50000" "14000,46000" "18000,44000" "20000,40000" "16000,50000" "12000,42000" "22000,46000"; do
echo "clients $N slices $SL: $(XL_EXP_POLY_SLICES=$SL python tools/sweep.py --clients $N --rates 5 --modes optimized --steps 200 2>&1 | grep optimized | awk '{print $5, $6, $10}')"
done; done
