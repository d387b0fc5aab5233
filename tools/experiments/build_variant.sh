#!/bin/bash
# tools/experiments/build_variant.sh NAME "EXTRA HIPCC FLAGS" -- a variant of libxlating_hip.so with xl_polyphase.hip compiled with extra
# flags, into sdr-server_amd/build/variants/libNAME.so (travels with gpurun; select with XL_LIBRARY_PATH)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/sdr-server_amd/csrc; B=$ROOT/sdr-server_amd/build; V=$B/variants; mkdir -p $V
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt"
hipcc $FLAGS $2 -c $C/xl_polyphase.hip -o $V/$1_polyphase.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $V/lib$1.so $B/xl_kernels.o $V/$1_polyphase.o $B/xl_filter.o $B/xl_batch.o $B/xl_sinks.o $B/xl_common.o $B/lpf.o $B/xl_taps.o $B/xl_wire.o -lm -lz -lpthread
echo built $V/lib$1.so
