#!/bin/bash
# tools/experiments/build_tune.sh -- sdr-server_amd/build/variants/libtune.so: the library with the engine AND the two-half mix kernels
# compiled -DXL_TUNING (launch traces: XL_EXP_POLY_TRACE, tools/r06_trace.py, tools/fwd_trace.py).  Select with XL_TESTING=1 XL_LIBRARY_PATH=...
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/sdr-server_amd/csrc; B=$ROOT/sdr-server_amd/build; V=$B/variants; mkdir -p $V
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -DXL_TUNING $1"
hipcc $FLAGS -c $C/xl_batch.cpp -o $V/tune_xl_batch.o &
hipcc $FLAGS -fno-slp-vectorize -c $C/xl_mixh.hip -o $V/tune_xl_mixh.o &
hipcc $FLAGS -fno-slp-vectorize -c $C/xl_mixh2.hip -o $V/tune_xl_mixh2.o &
wait
OBJS=""
for o in xl_kernels xl_polyphase xl_inv8 xl_inv32 xl_mixf32 xl_mixh xl_mixh2 xl_filter xl_batch xl_sinks xl_common lpf xl_taps xl_wire; do
  case $o in xl_batch|xl_mixh|xl_mixh2) OBJS="$OBJS $V/tune_$o.o";; *) OBJS="$OBJS $B/$o.o";; esac
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libtune.so $OBJS -lm -lz -lpthread
echo built $V/libtune.so
