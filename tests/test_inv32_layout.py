"""CPU test (-m "not gpu") of the 32 x 4 inverse launch's index bookkeeping (sdr-server_amd/csrc/xl_inv32_layout.h and the 32- / 4-point
register transforms of xl_fft16.h, the headers xlp_inverse32_kernel takes its indices from): compiled for the host and driven through
an emulation of a wave's lanes (tests/c/test_inv32_layout.cpp) -- every output of every column equal to IDFT_128 of its bins / 128
through the two exchange rounds, the tile loaded exactly once in whole 128-byte lines, the LDS accesses of the kernel free of bank
conflicts under the guide's per-instruction lane groups (phase reads: at most 2-way), the regions disjoint and sixteen waves' worth
of them within a CU's LDS."""
import os
import subprocess

import pytest

from conftest import ROOT

CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang (ext_vector_type)")
def test_inverse32_layout_data_flow_and_lds_banks(tmp_path):
    exe = str(tmp_path / "test_inv32_layout")
    r = subprocess.run([CLANG, "-std=c++17", "-O2", os.path.join(ROOT, "tests", "c", "test_inv32_layout.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "inv32 layout: ok" in r.stdout
