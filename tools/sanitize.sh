#!/bin/bash
# tools/sanitize.sh -- the memory-checker pass over the HOST code of the library (stand-in for the reference's valgrind runs,
# test/resources/run_tests.sh:8; no valgrind in this image).  Builds the host-side sources a second time with
#   -fsanitize=address,undefined   (lpf.c, xl_taps.c, xl_wire.c, xl_sinks.cpp; xl_grid.h through tests/test_grid.py's shim)
#   -fsanitize=thread              (xl_sinks.cpp: writer threads, bounded queues; + the C sources it is linked with)
# into sdr-server_amd/build/sanitize/libxlating_host_{asan,tsan}.so and drives them with the EXISTING CPU tests
# (tests/test_sinks.py, test_wire.py, test_grid.py, the lpf / tap-preparation tests of test_capi_boundary.py).  Symbols of
# the library that live in HIP translation units are stubbed with abort() (generated here from the real library's export
# list), so a test that strays onto the GPU path fails loudly.  Logs: profiles/r05_sanitize_{asan_ubsan,tsan}.txt.
# CPU only.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/sdr-server_amd/csrc
B=$ROOT/sdr-server_amd/build/sanitize
REAL=$ROOT/sdr-server_amd/lib/libxlating_hip.so
mkdir -p $B $ROOT/profiles
TESTS="tests/test_sinks.py tests/test_wire.py tests/test_grid.py tests/test_capi_boundary.py"
# (left out: tests of the GPU library's symbol table / device probe, and the one that calls create_frequency_xlating_filter)
KEXPR="not test_library_exists and not every_declared and not exported_list and not no_oracle and not fails_loudly and not rejects_empty"
build() {  # $1 = tag, $2 = sanitizer flags
  local tag=$1 flags=$2 d=$B/$1
  rm -rf $d
  mkdir -p $d
  for f in lpf xl_taps xl_wire; do gcc -std=c11 -O1 -g -fno-omit-frame-pointer -fno-fast-math -ffp-contract=off -fPIC $flags -c $CS/$f.c -o $d/$f.o || return 1; done
  g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fPIC $flags -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -c $CS/xl_sinks.cpp -o $d/xl_sinks.o || return 1
  # everything the real library exports and these objects do not define: abort() stubs (functions), a string (SIMD_STATUS)
  nm -D --defined-only $REAL | awk '$2=="T"{print $3}' | sort > $d/exported.txt
  nm --defined-only $d/*.o | awk '$2=="T"{print $3}' | sort -u > $d/defined.txt
  { echo '#include <stdlib.h>'; echo 'const char *SIMD_STATUS = "host-only sanitizer build";';
    comm -23 $d/exported.txt $d/defined.txt | grep -v '^_' | while read s; do echo "void $s(void) { abort(); }"; done; } > $d/stubs.c
  gcc -O0 -fPIC -c $d/stubs.c -o $d/stubs.o || return 1
  g++ -shared -fPIC $flags -o $B/libxlating_host_$tag.so $d/*.o -lm -lz -lpthread || return 1
}
run() {  # $1 = tag, $2 = runtime library, $3 = log, $4 = test files, $5... = env
  local tag=$1 rt=$2 log=$3 tests=$4; shift 4
  rm -f $B/report_$tag.*
  ( cd $ROOT && timeout 600 env "$@" LD_PRELOAD=$rt XL_TESTING=1 XL_LIBRARY_PATH=$B/libxlating_host_$tag.so python -m pytest $tests -q -x -p no:cacheprovider -k "$KEXPR" ) > $log.tmp 2>&1
  local rc=$?
  cat $B/report_$tag.* >> $log.tmp 2>/dev/null  # (log_path: pytest captures the tests' stderr)
  local TESTS=$tests
  { echo "# tools/sanitize.sh: $tag build of lpf.c xl_taps.c xl_wire.c xl_sinks.cpp (+ xl_grid.h shim), $(gcc --version | head -1)";
    echo "# command: LD_PRELOAD=$(basename $rt) XL_TESTING=1 XL_LIBRARY_PATH=libxlating_host_$tag.so pytest $TESTS -k \"$KEXPR\"";
    echo "# exit code $rc; sanitizer reports below (none = clean)"; grep -E "ERROR: (Address|Thread|Leak)Sanitizer|runtime error:|WARNING: ThreadSanitizer|SUMMARY:" $log.tmp | sort | uniq -c | head -50;
    echo "# pytest tail:"; tail -4 $log.tmp; } > $log
  rm -f $log.tmp
  return $rc
}
rc=0
build asan "-fsanitize=address,undefined -fno-sanitize-recover=undefined" || { echo "asan build failed"; exit 1; }
run asan "$(gcc -print-file-name=libasan.so)" $ROOT/profiles/r05_sanitize_asan_ubsan.txt "$TESTS" \
    ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:log_path=$B/report_asan UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path=$B/report_asan \
    XL_SANITIZE_CFLAGS="-fsanitize=address,undefined" || rc=1
build tsan "-fsanitize=thread" || { echo "tsan build failed"; exit 1; }
# (tests/test_grid.py forks gcc, which does not return under a preloaded libtsan; xl_grid.h is single-threaded integer code)
run tsan "$(gcc -print-file-name=libtsan.so)" $ROOT/profiles/r05_sanitize_tsan.txt "tests/test_sinks.py tests/test_wire.py tests/test_capi_boundary.py" \
    TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:log_path=$B/report_tsan XL_SANITIZE_CFLAGS="" || rc=1
cat $ROOT/profiles/r05_sanitize_asan_ubsan.txt $ROOT/profiles/r05_sanitize_tsan.txt
exit $rc
