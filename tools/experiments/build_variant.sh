#!/bin/bash
# tools/experiments/build_variant.sh NAME "EXTRA HIPCC FLAGS" [SOURCE] -- a variant of libxlating_hip.so with one kernel file (default
# xl_polyphase.hip; e.g. xl_inv8.hip) compiled with extra flags, into sdr-server_amd/build/variants/libNAME.so (travels with gpurun;
# select with XL_TESTING=1 XL_LIBRARY_PATH=...): several hypotheses per GPU call
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/sdr-server_amd/csrc; B=$ROOT/sdr-server_amd/build; V=$B/variants; mkdir -p $V
SRC=${3:-xl_polyphase.hip}; STEM=${SRC%.*}   # (.hip or .cpp: e.g. xl_batch.cpp with -DXL_TUNING for the launch traces)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt"
case $STEM in xl_mixh|xl_mixh2|xl_mixf32) FLAGS="$FLAGS -fno-slp-vectorize";; esac   # (csrc/Makefile: MIX_FLAGS)
hipcc $FLAGS $2 -c $C/$SRC -o $V/$1_$STEM.o
OBJS=""
for o in xl_kernels xl_polyphase xl_inv8 xl_inv32 xl_mixf32 xl_mixh xl_mixh2 xl_filter xl_batch xl_sinks xl_common lpf xl_taps xl_wire; do
  if [ "$o" = "$STEM" ]; then OBJS="$OBJS $V/$1_$STEM.o"; else OBJS="$OBJS $B/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $V/lib$1.so $OBJS -lm -lz -lpthread
echo built $V/lib$1.so
