"""CPU test (-m "not gpu") of the matrix-core mix's index bookkeeping (sdr-server_amd/csrc/xl_mix_layout.h, the header
xlp_mix_mfma_kernel and xlp_tables_h_kernel take their slot / lane / register indices from): compiled for the host and
driven through an emulation of the matrix instruction's operand and result maps (tests/c/test_mix_layout.cpp) -- every
operand slot written exactly where the products look for it, every result register stored as the right (segment, column)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CXX = shutil.which("g++") or "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CXX), reason="needs a C++ compiler")
def test_matrix_core_mix_layout_against_plain_complex_sums(tmp_path):
    exe = str(tmp_path / "test_mix_layout")
    r = subprocess.run([CXX, "-std=c++17", "-O1", os.path.join(ROOT, "tests", "c", "test_mix_layout.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "matrix-core mix layout: ok" in r.stdout

