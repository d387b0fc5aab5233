"""CPU test (-m "not gpu") of the fused mix + inverse launch's index bookkeeping (sdr-server_amd/csrc/xl_fused_layout.h and the
32-point register transform of xl_fft64.h, the headers xlp_forward_h_kernel / xlp_tables_h16_kernel / xlp_fused_kernel take
their slot, lane, register and exchange-buffer indices from): compiled for the host and driven through an emulation of the
16 x 16 x 32 matrix instruction's operand and result maps, the 4 x 32 split of the inverse transform and the exchange between
the four waves (tests/c/test_fused_layout.cpp) -- every operand slot written exactly where the products look for it, every
output of every (segment, client column) produced exactly once and equal to IDFT_128(sum_b X R)."""
import os
import subprocess

import pytest

from conftest import ROOT

CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang (ext_vector_type)")
def test_fused_launch_layout_against_plain_complex_arithmetic(tmp_path):
    exe = str(tmp_path / "test_fused_layout")
    r = subprocess.run([CLANG, "-std=c++17", "-O2", os.path.join(ROOT, "tests", "c", "test_fused_layout.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fused layout: ok" in r.stdout
