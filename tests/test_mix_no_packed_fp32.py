"""CPU test (-m "not gpu"): the mix kernels' code objects contain no PACKED FP32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32).
A wave that shares a SIMD with matrix instructions loses lanes 48..63 of a packed-FP32 result once in a while on this chip (the NCO role's
packed recurrence steps, rounds 3-4: DESIGN_HISTORY.md 3.6; a mix kernel's own `sum x scale` epilogue, round 6: profiles/
r06_mix_wide_kmajor_wrong_sums.txt (11)); single-lane-width FP32 is immune.  The mix kernels form their float32 products and sums through
inline-asm helpers (csrc/xl_poly_dev.h: xl_mul_s / xl_add_s / xl_sub_s) the vectoriser cannot re-pack; this compiles them for gfx950 and
looks (hipcc cross-compiles without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "--cuda-device-only", "-S"]


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
@pytest.mark.parametrize("src", ["xl_mixh2.hip", "xl_mixf32.hip", "xl_polyphase.hip"])
def test_mix_kernels_issue_no_packed_fp32(src, tmp_path):
    out = str(tmp_path / "k.s")
    r = subprocess.run(["hipcc"] + FLAGS + [os.path.join(ROOT, "sdr-server_amd", "csrc", src), "-o", out], capture_output=True, text=True)
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
    assert not bad, bad[:5]
    # ... and they DO issue matrix instructions (the premise)
    assert re.search(r"v_mfma_f32_32x32x(16_f16|2_f32)", open(out).read())
