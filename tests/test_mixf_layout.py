"""CPU test (-m "not gpu") of the index bookkeeping of the mix launch on the matrix cores with float32 operands
(sdr-server_amd/csrc/xl_mixf_layout.h, the header xlp_mix_f32_kernel and xlp_tables_f_kernel take their slot / lane / register
indices from): compiled for the host and driven through an emulation of v_mfma_f32_32x32x2_f32's operand and result maps
(tests/c/test_mixf_layout.cpp) -- every B operand written exactly where the products look for it, the A operand read straight
out of the forward launch's image rows, every result register stored as the right (segment, column)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CXX = shutil.which("g++") or "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CXX), reason="needs a C++ compiler")
def test_float32_matrix_core_mix_layout_against_plain_complex_sums(tmp_path):
    exe = str(tmp_path / "test_mixf_layout")
    r = subprocess.run([CXX, "-std=c++17", "-O1", os.path.join(ROOT, "tests", "c", "test_mixf_layout.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "float32 matrix-core mix layout: ok" in r.stdout
