"""CPU test (-m "not gpu") of the "8 lanes per column" inverse launch's index bookkeeping (sdr-server_amd/csrc/xl_inv8_layout.h and
the 16- / 8-point register transforms of xl_fft64.h, the headers xlp_inverse8_kernel takes its indices from): compiled for the host and driven through an emulation of a wave's lanes (tests/c/test_inv8_layout.cpp) -- every
output of every column equal to IDFT_128 of its bins, the tile loaded exactly once, and every
LDS access of the kernel free of bank conflicts under the guide's per-instruction lane groups."""
import os
import subprocess

import pytest

from conftest import ROOT

CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang (ext_vector_type)")
def test_inverse8_layout_data_flow_and_lds_banks(tmp_path):
    exe = str(tmp_path / "test_inv8_layout")
    r = subprocess.run([CLANG, "-std=c++17", "-O2", os.path.join(ROOT, "tests", "c", "test_inv8_layout.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "inv8 layout: ok" in r.stdout
