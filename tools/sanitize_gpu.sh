#!/bin/bash
# tools/sanitize_gpu.sh build | run <out-dir> -- AddressSanitizer + UBSan over the HOST code of the GPU engine (xl_batch.cpp, xl_filter.cpp,
# xl_multi.cpp, xl_common.cpp, xl_sinks.cpp: plans, rings of phase tables and events, row allocator, stream bookkeeping) on a real GPU:
# the part of the library tools/sanitize.sh (CPU only) cannot reach.  Stand-in for the reference's valgrind runs
# (test/resources/run_tests.sh:8).
#   build   (build container) the engine's .cpp files a second time with g++ -fsanitize=address,undefined (they hold no kernels: plain
#           C++ over the HIP headers), linked with the normal kernel objects into sdr-server_amd/build/sanitize_gpu/libxlating_hip_asan.so
#           (travels with gpurun; the plain build under sdr-server_amd/build/ must be current)
#   run     (GPU box) drives it with GPU tests that stay off torch (host-path calls): churn with a join and a leave per block, CU
#           reservation (also in rounds: 4096 clients), options, re-plans, groups of blocks, config 5, the drop-in filter's create / process / destroy cycles
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/sdr-server_amd/csrc; B=$ROOT/sdr-server_amd/build; S=$B/sanitize_gpu
RT=$(gcc -print-file-name=libasan.so)   # (gcc's runtime: ROCm clang's ASan intercepts the HSA allocator and wants a GPU-side set-up of its own)
case "${1:-}" in
build)
  mkdir -p $S
  FLAGS="-std=c++17 -O1 -g -fno-omit-frame-pointer -ffp-contract=off -fno-fast-math -fPIC -Wall -Wno-unused-function -Wno-unknown-pragmas -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
  for f in xl_batch xl_filter xl_common xl_sinks; do g++ $FLAGS -c $CS/$f.cpp -o $S/$f.o || exit 1; done
  OBJS="$B/xl_kernels.o $B/xl_polyphase.o $B/xl_inv8.o $B/xl_inv32.o $B/xl_mixf32.o $B/xl_mixh.o $B/xl_mixh2.o $S/xl_filter.o $S/xl_batch.o $S/xl_sinks.o $S/xl_common.o $B/lpf.o $B/xl_taps.o $B/xl_wire.o"
  g++ -shared -fPIC -fsanitize=address,undefined -o $S/libxlating_hip_asan.so $OBJS -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lm -lz -lpthread || exit 1
  rm -f $S/*.o
  echo built $S/libxlating_hip_asan.so ;;
run)
  OUT=${2:-$ROOT/gpurun_out/sanitize}; mkdir -p $OUT; rm -f $S/report.*
  K="churn or expected_clients or set_option or describe_after or size_rule or group_of_blocks_polyphase or config5_cf32_10msps_all_clients or polyphase_forced_server_default or chain_launch_covers or group_4096_clients_sampled"
  ( cd $ROOT && timeout 1200 env LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:protect_shadow_gap=0:log_path=$S/report UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$S/report \
      XL_TESTING=1 XL_LIBRARY_PATH=$S/libxlating_hip_asan.so python -m pytest tests/test_batch_gpu.py tests/test_c_dropin.py -m gpu -q -p no:cacheprovider -k "$K" ) > $OUT/asan_gpu.tmp 2>&1
  rc=$?
  cat $S/report.* > $OUT/asan_gpu_reports.txt 2>/dev/null
  { echo "# tools/sanitize_gpu.sh: ASan + UBSan build of xl_batch.cpp xl_filter.cpp xl_common.cpp xl_sinks.cpp (g++ -fsanitize=address,undefined), kernels uninstrumented, on $(rocminfo 2>/dev/null | grep -m1 gfx9 | xargs)";
    echo "# command: LD_PRELOAD=$(basename $RT) XL_TESTING=1 XL_LIBRARY_PATH=libxlating_hip_asan.so pytest tests/test_batch_gpu.py tests/test_c_dropin.py -m gpu -k \"$K\"";
    echo "# exit code $rc; reports with a frame in this library (none = clean; the HIP runtime's and Python's own are not ours to fix):";
    grep -c "ERROR: AddressSanitizer\|runtime error:" $OUT/asan_gpu_reports.txt 2>/dev/null | sed 's/^/# reports in total: /';
    grep -B2 -A12 "ERROR: AddressSanitizer\|runtime error:" $OUT/asan_gpu_reports.txt 2>/dev/null | grep -B6 -A6 "xl_batch\|xl_filter\|xl_common\|xl_sinks\|libxlating" | head -80;
    echo "# pytest tail:"; tail -4 $OUT/asan_gpu.tmp; } > $OUT/sanitize_gpu_asan_ubsan.txt
  cat $OUT/sanitize_gpu_asan_ubsan.txt
  exit $rc ;;
*) echo "usage: $0 build | run <out-dir>"; exit 2 ;;
esac
