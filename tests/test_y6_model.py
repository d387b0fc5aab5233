"""CPU test (-m "not gpu") of the 48-bit mixed-spectra format (sdr-server_amd/csrc/xl_y6.h, shared by xlp_mix_mfma_kernel and
xlp_inverse_kernel): the header compiled for the host round-trips random, extreme and tiny values; the error is what the header
promises (at most 2^-20 of the larger component), the column factor is applied exactly, and -- a model of a whole inverse transform --
quantising every bin of random spectra adds far less than the 1e-5 bar to the outputs."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CC = shutil.which("gcc")

SRC = r'''
#include "%s/sdr-server_amd/csrc/xl_y6.h"
void roundtrip(const float *re, const float *im, int n, int kexp, float *ore, float *oim) {
  for (int i = 0; i < n; ++i) {
    uint32_t lo, hi;
    xly6_encode(re[i], im[i], &lo, &hi);
    if (hi >> 16) { ore[i] = oim[i] = NAN; continue; }  /* must fit 16 bits */
    xly6_decode(lo, hi, kexp, &ore[i], &oim[i]);
  }
}
''' % ROOT


@pytest.mark.skipif(CC is None, reason="needs gcc")
def test_y6_round_trip_and_error_bound(tmp_path):
    c = tmp_path / "y6.c"
    c.write_text(SRC)
    so = str(tmp_path / "y6.so")
    r = subprocess.run([CC, "-O2", "-shared", "-fPIC", "-o", so, str(c), "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = C.CDLL(so)
    fp = C.POINTER(C.c_float)
    lib.roundtrip.argtypes = [fp, fp, C.c_int, C.c_int, fp, fp]

    def rt(re, im, kexp=0):
        re = np.ascontiguousarray(re, np.float32)
        im = np.ascontiguousarray(im, np.float32)
        ore, oim = np.empty_like(re), np.empty_like(im)
        lib.roundtrip(re.ctypes.data_as(fp), im.ctypes.data_as(fp), re.size, kexp, ore.ctypes.data_as(fp), oim.ctypes.data_as(fp))
        return ore, oim

    rng = np.random.default_rng(5)
    # magnitudes from 2^-30 to 2^35 (the mix's sums stay below 2^35), components of very different size
    mag = np.exp2(rng.uniform(-30, 35, 200000)).astype(np.float32)
    re = (mag * rng.uniform(-1, 1, mag.size)).astype(np.float32)
    im = (mag * rng.uniform(-1, 1, mag.size) * np.exp2(rng.integers(-24, 1, mag.size))).astype(np.float32)
    ore, oim = rt(re, im)
    big = np.maximum(np.abs(re), np.abs(im)).astype(np.float64)
    err = np.maximum(np.abs(ore.astype(np.float64) - re), np.abs(oim.astype(np.float64) - im))
    ok = big >= 2.0 ** -27  # (below: the exponent field is at its end, the step is absolute: 2^-46)
    assert np.all(err[ok] <= big[ok] * 2.0 ** -20 * (1 + 1e-9)), float((err[ok] / big[ok]).max())
    assert np.all(err[~ok] <= 2.0 ** -47)
    # exact cases: zero, powers of two, the largest sums, values that round up to the clamp
    re0 = np.array([0.0, 1.0, -2.0 ** 34, 3.0e10, 2.0 ** 20 - 0.25, -(2.0 ** 20 - 0.25), 1e-30], np.float32)
    im0 = np.array([0.0, -0.5, 2.0 ** 34, -3.1e10, 2.0 ** 20 - 0.25, 5.0, -1e-30], np.float32)
    o1, o2 = rt(re0, im0)
    assert o1[0] == 0 and o2[0] == 0 and o1[1] == 1 and o2[1] == -0.5 and o1[2] == -2.0 ** 34 and o2[2] == 2.0 ** 34
    assert abs(o1[3] - 3.0e10) <= 3.0e10 * 2.0 ** -21 and abs(o1[4] - re0[4]) <= 1.0 and o1[6] == 0 and o2[6] == 0
    # the column factor is an exponent offset: exact
    a1, b1 = rt(re[:1000], im[:1000], kexp=-23)
    a0, b0 = rt(re[:1000], im[:1000])
    assert np.array_equal(a1, np.ldexp(a0, -23).astype(np.float32)) and np.array_equal(b1, np.ldexp(b0, -23).astype(np.float32))
    # a whole segment: 128 bins of a noise-like spectrum quantised, inverse transform, against the unquantised transform
    worst = 0.0
    for t in range(20):
        Y = (rng.normal(size=128) + 1j * rng.normal(size=128)) * np.exp2(rng.uniform(18, 30))
        Y *= np.exp2(rng.uniform(-8, 0, 128))  # (bins of different size, like a filtered spectrum)
        qr, qi = rt(Y.real, Y.imag)
        y0 = np.fft.ifft(Y.real.astype(np.float32).astype(np.float64) + 1j * Y.imag.astype(np.float32).astype(np.float64))
        y1 = np.fft.ifft(qr.astype(np.float64) + 1j * qi.astype(np.float64))
        worst = max(worst, float(np.abs(y1 - y0).max() / np.abs(y0).max()))
    assert worst < 1e-6, worst  # (measured 6.4e-7 in the worst of these trials: a sixteenth of the 1e-5 bar, on top of 1.6e-6)
