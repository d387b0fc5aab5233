"""A plain C caller (tests/c/dropin_demo.c, written like the reference's own callers) compiled with gcc against
include/*.h and linked to libxlating_hip.so: it must build without any HIP header (CPU test) and, on the GPU box,
print exactly the oracle's samples (gpu test)."""
import os
import subprocess

import numpy as np
import pytest

import sdr_server_amd as xl
from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "c", "dropin_demo.c")
EXE = os.path.join(ROOT, "sdr-server_amd", "build", "dropin_demo")


def build_demo():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.dirname(xl.library_path())
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
           "-L", libdir, "-lxlating_hip", f"-Wl,-rpath,{libdir}", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_c_caller_compiles_and_links_against_the_dropin_headers():
    build_demo()
    out = subprocess.run(["nm", "-u", EXE], capture_output=True, text=True).stdout
    for sym in ("create_low_pass_filter", "create_frequency_xlating_filter", "process_native_cu8_cf32", "destroy_xlating"):
        assert sym in out, sym
    assert "SIMD_STATUS" in subprocess.run(["nm", EXE], capture_output=True, text=True).stdout  # data symbol (copy reloc)
    assert "hip" not in out.lower()  # the caller never touches HIP


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["native", "optimized"])
def test_c_caller_output_matches_oracle(variant):
    from pyoracle import Oracle

    exe = EXE if os.path.exists(EXE) else build_demo()
    fs, rate, tw, fc, nbytes, ncalls = 2016000, 48000, 9600, -12000, 100002, 3
    r = subprocess.run([exe, variant, str(fs), str(rate), str(tw), str(fc), str(nbytes), str(ncalls)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "HIP gfx950" in r.stderr
    taps = Oracle.lpf(1.0, fs, rate // 2, tw)[1]
    o = Oracle(fs // rate, taps, fc, fs, nbytes)
    lines = r.stdout.strip().splitlines()
    assert len(lines) == ncalls
    for call, line in enumerate(lines):
        parts = line.split()
        x = ((call * nbytes + np.arange(nbytes)) & 0xFF).astype(np.uint8)
        want = o.process("cu8", x)
        assert int(parts[0]) == len(want)
        got = np.array([int(h, 16) for h in parts[1:]], dtype=np.uint64)
        got = np.stack([(got >> np.uint64(32)).astype(np.uint32), (got & np.uint64(0xFFFFFFFF)).astype(np.uint32)], axis=1)
        got = got.reshape(-1).view(np.float32).view(np.complex64)
        if variant == "native":
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), call
        else:
            assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), call


# ---- the reference's OWN callers (test/perf_xlating.c, test/test_xlating.c), compiled where they lie against the
# reference's own headers and linked to libxlating_hip.so by tests/c/Makefile -> tests/c/_refbin/ (shipped to the GPU box)
REFBIN = os.path.join(ROOT, "tests", "c", "_refbin")
API = ["create_low_pass_filter", "create_frequency_xlating_filter", "process_native_cu8_cf32",
       "process_optimized_cu8_cf32", "process_native_cu8_cs16", "process_optimized_cu8_cs16"]


@pytest.mark.skipif(not os.path.exists("/root/reference/test/perf_xlating.c"), reason="build container only (needs /root/reference)")
def test_reference_callers_link_against_the_library():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    perf, unit = os.path.join(REFBIN, "perf_xlating_hip"), os.path.join(REFBIN, "test_xlating_hip")
    und = subprocess.run(["nm", "-u", perf], capture_output=True, text=True).stdout
    for sym in API:
        assert sym in und, sym  # unresolved in the reference's object, resolved by libxlating_hip.so at load time
    assert "SIMD_STATUS" in subprocess.run(["nm", perf], capture_output=True, text=True).stdout
    und = subprocess.run(["nm", "-u", unit], capture_output=True, text=True).stdout
    for sym in ("create_frequency_xlating_filter", "process_native_cu8_cf32", "process_native_cu8_cs16", "destroy_xlating"):
        assert sym in und, sym
    for exe in (perf, unit):
        assert "libxlating_hip.so" in subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    defined = subprocess.run(["nm", "-D", "--defined-only", xl.library_path()], capture_output=True, text=True).stdout
    for sym in API + ["SIMD_STATUS", "destroy_xlating"]:
        assert f" {sym}" in defined, sym


@pytest.mark.gpu
def test_reference_unit_test_passes_against_the_library():
    """test/test_xlating.c (Unity): its three tests with the reference's expected arrays, run unmodified on the GPU."""
    exe = os.path.join(REFBIN, "test_xlating_hip")
    if not os.path.exists(exe):
        pytest.skip("tests/c/_refbin not built (needs the build container)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "3 Tests 0 Failures 0 Ignored" in r.stdout, r.stdout[-2000:]


@pytest.mark.gpu
def test_reference_perf_program_runs_against_the_library():
    """test/perf_xlating.c: 4 x 1000 calls of 200000 bytes through the drop-in API; prints SIMD_STATUS and four timings."""
    exe = os.path.join(REFBIN, "perf_xlating_hip")
    if not os.path.exists(exe):
        pytest.skip("tests/c/_refbin not built (needs the build container)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert "SIMD optimization: HIP gfx950" in r.stdout
    times = [float(l.split(":")[1].split()[0]) for l in r.stdout.splitlines() if "seconds" in l]
    assert len(times) == 4 and all(0.0 < t < 0.01 for t in times), r.stdout
    print(r.stdout)


@pytest.mark.gpu
def test_create_process_destroy_cycle_is_leak_clean():
    """The reference runs every test under valgrind (test/resources/run_tests.sh:8).  Stand-in here: 1000 cycles of
    create / process (both families) / destroy of the drop-in filter and 200 of the batch engine (plan, polyphase images,
    group call, fetch) leave the device's free memory and the process's resident set flat."""
    import psutil
    import torch

    proc = psutil.Process()
    code, taps = xl.create_low_pass_filter(1.0, 2016000, 24000, 9600)
    x = ((np.arange(20000) * 7) & 0xFF).astype(np.uint8)

    def cycle_filter():
        f = xl.XlatingFilter(42, taps, -12000, 2016000, 20000)
        f.process("native", "cu8", "cf32", x)
        f.process("optimized", "cu8", "cs16", x)
        f.close()

    def cycle_engine():
        e = xl.BatchEngine(2016000, "cu8", 20000, group_blocks=2)
        e.set_option("polyphase", 1)
        for c in range(9):
            e.add_client(42, taps, 1000 * c)
        e.process_host(x, "optimized")
        e.process_host_group(np.concatenate([x, x]), 2, "optimized")
        e.fetch()
        e.remove_client(3)
        e.process_host(x, "native")
        e.close()

    for _ in range(30):  # warm up allocator pools, code objects, the HIP runtime's own caches
        cycle_filter()
    for _ in range(10):
        cycle_engine()
    torch.cuda.synchronize()
    free0, rss0 = torch.cuda.mem_get_info()[0], proc.memory_info().rss
    for _ in range(1000):
        cycle_filter()
    for _ in range(200):
        cycle_engine()
    torch.cuda.synchronize()
    free1, rss1 = torch.cuda.mem_get_info()[0], proc.memory_info().rss
    assert free0 - free1 <= 8 << 20, f"device memory leaked: {free0 - free1} bytes"
    assert rss1 - rss0 <= 48 << 20, f"host memory leaked: {rss1 - rss0} bytes"


# ---- latency of the drop-in call as dsp_worker.c:49-86 makes it: one thread, one filter, one synchronous call per block
LAT_SRC = os.path.join(ROOT, "tests", "c", "dropin_latency.c")
LAT_EXE = os.path.join(ROOT, "sdr-server_amd", "build", "dropin_latency")


def build_latency():
    os.makedirs(os.path.dirname(LAT_EXE), exist_ok=True)
    libdir = os.path.dirname(xl.library_path())
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), LAT_SRC, "-o", LAT_EXE,
           "-L", libdir, "-lxlating_hip", f"-Wl,-rpath,{libdir}", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return LAT_EXE


def test_latency_harness_compiles():
    build_latency()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["native", "optimized"])
def test_ten_thousand_consecutive_calls_stay_under_a_millisecond(variant):
    """Round 2 reported "about one call in a thousand takes 1-35 ms" for the drop-in filter.  Root cause (round 3,
    tools/dropin_python_stall.py): the MEASURING process -- CPython's cyclic garbage collector stopping the interpreter in
    the middle of the timed wrapper call (37 ms once in 10 000 with gc enabled, never with gc disabled, never around the bare
    ctypes call).  From C, the way dsp_worker.c calls the filter, 10 000 consecutive 262144-byte calls show no call over
    0.2 ms.  This test keeps it that way: max < 1 ms (one retry: the box is shared with other jobs' host threads)."""
    import json

    exe = build_latency()
    last = None
    for attempt in range(2):
        r = subprocess.run([exe, variant, "10000"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        last = json.loads(r.stdout.strip().splitlines()[-1])
        if last["calls_over_1ms"] == 0:
            break
    assert last["calls_over_1ms"] == 0 and last["max_us"] < 1000.0, last
    assert last["median_us"] < 120.0, last
