#!/bin/bash
# tools/r06_shapes.sh <outdir> -- a `custom:` step of tools/gpu_r06.sh: the plan's size rules on shapes they were NOT fitted to (VERDICT r5 weak 9,
# ADVICE r5 #3).  1024 clients x 8 blocks, cu8 input at 48 kHz x D, transition 48 kHz / rate (taps ~ 2.4 D rate): the engine's own choice
# (polyphase = -1: auto) against the polyphase path forced (1) and the direct kernel forced (0); and the transform length 128 / 256 forced.
OUT=$1
for D in 8 21 42 100 200; do
  for rate in 1 5 24; do
    for p in -1 1 0; do
      timeout 300 python tools/group_sweep.py --decimations $D --rate $rate --clients 1024 --groups 8 --modes optimized --blocks 160 --opt polyphase=$p 2>&1 | grep "^optimized" | sed "s/^/D=$D rate=$rate polyphase=$p  /"
    done
    for m in 128 256; do
      timeout 300 python tools/group_sweep.py --decimations $D --rate $rate --clients 1024 --groups 8 --modes optimized --blocks 160 --opt polyphase=1 --m $m 2>&1 | grep "^optimized" | sed "s/^/D=$D rate=$rate polyphase=1 M=$m  /"
    done
  done
done | tee $OUT/shapes_1024clients.txt
for D in 21 100; do
  for n in 64 256; do
    for p in -1 1 0; do
      timeout 300 python tools/group_sweep.py --decimations $D --rate 5 --clients $n --groups 8 --modes optimized --blocks 160 --opt polyphase=$p 2>&1 | grep "^optimized" | sed "s/^/D=$D rate=5 polyphase=$p  /"
    done
  done
done | tee $OUT/shapes_small_classes.txt
