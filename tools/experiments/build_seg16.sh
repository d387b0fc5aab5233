#!/bin/bash
# tools/experiments/build_seg16.sh -- libxlating_hip.so with 16 segments per pass of the mix launches (XLP_SEG = 16: all 32 rows of a
# matrix instruction used) into sdr-server_amd/build/variants/libseg16.so; the packed-FMA mix kernel is written for 14 and is left out
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/sdr-server_amd/csrc; B=$ROOT/sdr-server_amd/build; V=$B/variants; mkdir -p $V
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -DXLP_SEG=16u $1"
for f in xl_polyphase.hip xl_mixf32.hip xl_batch.cpp; do hipcc $FLAGS -c $C/$f -o $V/seg16_${f%.*}.o & done; wait
OBJS=""
for o in xl_kernels xl_polyphase xl_fused xl_inv8 xl_mixf32 xl_filter xl_batch xl_sinks xl_common lpf xl_taps xl_wire; do
  if [ -f $V/seg16_$o.o ]; then OBJS="$OBJS $V/seg16_$o.o"; else OBJS="$OBJS $B/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libseg16.so $OBJS -lm -lz -lpthread
echo built $V/libseg16.so
