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
