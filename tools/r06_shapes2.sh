#!/bin/bash
# tools/r06_shapes2.sh <outdir>: the crossover of classes on the STREAMED float32 mix (D > 112) against the direct kernel at small client counts
OUT=$1
for D in 128 200 400; do
  for rate in 1 2 5; do
    for n in 32 64 128 256 1024; do
      for p in 1 0; do
        timeout 300 python tools/group_sweep.py --decimations $D --rate $rate --clients $n --groups 8 --modes optimized --blocks 160 --opt polyphase=$p 2>&1 | grep "^optimized" | sed "s/^/D=$D rate=$rate polyphase=$p  /"
      done
    done
  done
done | tee $OUT/shapes_streamed_mix.txt
