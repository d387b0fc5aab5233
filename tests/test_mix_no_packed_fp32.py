"""CPU test (-m "not gpu"): the mix kernels' code objects contain no PACKED FP32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32).
A wave that shares a SIMD with matrix instructions loses lanes 48..63 of a packed-FP32 result once in a while on this chip (the NCO role's
packed recurrence steps, rounds 3-4: DESIGN_HISTORY.md 3.6; a mix kernel's own `sum x scale` epilogue, round 6: profiles/
r06_mix_wide_kmajor_wrong_sums.txt (11)); single-lane-width FP32 is immune.  The three files that hold mix kernels are compiled without the
SLP vectoriser (csrc/Makefile: MIX_FLAGS = -fno-slp-vectorize) -- which is what forms packed FP32 from scalar code; this compiles them for
gfx950 with the Makefile's flag and looks (hipcc cross-compiles without a GPU), and checks that the Makefile applies the flag to them."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "--cuda-device-only", "-S"]


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
@pytest.mark.parametrize("src", ["xl_mixh.hip", "xl_mixh2.hip", "xl_mixf32.hip"])
def test_mix_kernels_issue_no_packed_fp32(src, tmp_path):
    mk = open(os.path.join(ROOT, "sdr-server_amd", "csrc", "Makefile")).read()
    m = re.search(r"^MIX_FLAGS\s*:=\s*(.+)$", mk, re.M)
    assert m and "-fno-slp-vectorize" in m.group(1)
    rule = re.search(r"^(.*): HIPFLAGS \+= \$\(MIX_FLAGS\)$", mk, re.M)
    assert rule and ("$(BUILD)/" + src.replace(".hip", ".o")) in rule.group(1), "the Makefile must compile %s with MIX_FLAGS" % src
    out = str(tmp_path / "k.s")
    r = subprocess.run(["hipcc"] + FLAGS + m.group(1).split() + [os.path.join(ROOT, "sdr-server_amd", "csrc", src), "-o", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    kernel, mix_kernels, bad = None, set(), []
    for line in open(out):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            kernel = m.group(1)
            if "xlp_mix" in kernel:
                mix_kernels.add(kernel)
        elif kernel and "xlp_mix" in kernel and re.match(r"\s*v_pk_(mul|add|fma)_f32", line):
            bad.append((kernel, line.strip()))
    assert mix_kernels, "no mix kernel found in " + src  # (every one of the three files holds one)
    # no kernel that issues matrix instructions lives in any OTHER .hip file (they would be compiled with the vectoriser)
    for other in os.listdir(os.path.join(ROOT, "sdr-server_amd", "csrc")):
        if other.endswith(".hip") and other not in ("xl_mixh.hip", "xl_mixh2.hip", "xl_mixf32.hip"):
            assert "__builtin_amdgcn_mfma" not in open(os.path.join(ROOT, "sdr-server_amd", "csrc", other)).read(), other
    assert not bad, bad[:5]
    # ... and they DO issue matrix instructions (the premise)
    assert re.search(r"v_mfma_f32_32x32x(16_f16|2_f32)", open(out).read())
