"""CPU test (-m "not gpu") of the register FFT the inverse launch of the polyphase path runs
(sdr-server_amd/csrc/xl_fft64.h): the same header compiled for the host -- a two-"lane" emulation of the in-place
64-point transform, the pair exchange and the output slot map -- against a double-precision DFT (tests/c/test_fft64.cpp)."""
import os
import subprocess

import pytest

from conftest import ROOT

CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang (ext_vector_type)")
def test_register_fft_matches_a_double_precision_dft(tmp_path):
    exe = str(tmp_path / "test_fft64")
    r = subprocess.run([CLANG, "-std=c++17", "-O1", os.path.join(ROOT, "tests", "c", "test_fft64.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max relative error" in r.stdout
