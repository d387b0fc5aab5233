"""CPU model of the matrix-core mix's arithmetic (sdr-server_amd/csrc/xl_polyphase.hip, xlp_mix_mfma_kernel): every float32
operand scaled by a power of two and carried as TWO halves, three half x half products per complex-product term, FP32
accumulation.  The claim the kernel's header makes -- no worse than the FP32 FMA chain of xlp_mix_kernel, far inside the
1e-5 bar -- is pinned here on the server-default shape (D = 42, 505 taps, M = 128) without a GPU; the GPU tests then check
the kernel itself against the oracle."""
import numpy as np
import pytest

D, T, M = 42, 505, 128
A = -(-T // D)
V = M - A + 1


def _split(v, scale):
    v = (np.asarray(v, np.float32) * np.float32(scale)).astype(np.float32)
    h1 = v.astype(np.float16)
    h2 = (v - h1.astype(np.float32)).astype(np.float16)
    return h1.astype(np.float64), h2.astype(np.float64)


def _operands(seed, nseg=24):
    rng = np.random.default_rng(seed)
    n = np.arange(T) - (T - 1) / 2
    h = np.sinc(n * 0.9 / D) * np.hamming(T)
    h /= h.sum()
    r = (h * np.exp(2j * np.pi * 0.137 * np.arange(T))).astype(np.complex64)
    rb = np.zeros((D, A), np.complex128)
    for i in range(T):
        rb[i % D, i // D] = r[i]
    R = np.stack([np.fft.ifft(np.concatenate([rb[b], np.zeros(M - A)])) * M for b in range(D)]).astype(np.complex64)  # sum_a r_b[a] e^{+2 pi j a m / M}
    N = (nseg * V + M) * D + T
    x = (((rng.integers(0, 256, N) - 127.5) / 128) + 1j * ((rng.integers(0, 256, N) - 127.5) / 128)).astype(np.complex64)
    X = np.stack([[np.fft.fft(x[(s * V + np.arange(M)) * D + b].astype(np.complex128)) for b in range(D)] for s in range(nseg)]).astype(np.complex64)
    L = max(np.abs(rb[b]).sum() for b in range(D))
    return X, R, float(L)


def _err(Y, Yex):
    y, ye = np.fft.ifft(Y, axis=1)[:, :V], np.fft.ifft(Yex, axis=1)[:, :V]
    return np.abs(y - ye).max() / np.abs(ye).max()


def _mix_fp32_chain(X, R):
    ar = np.zeros(X.shape[::2], np.float32)
    ai = np.zeros_like(ar)
    for b in range(D):
        xr, xi, rr, ri = X[:, b].real.astype(np.float64), X[:, b].imag.astype(np.float64), R[b].real.astype(np.float64), R[b].imag.astype(np.float64)
        ar = (ar + rr * xr).astype(np.float32)  # (one rounding per FMA)
        ai = (ai + rr * xi).astype(np.float32)
        ar = (ar - ri * xi).astype(np.float32)
        ai = (ai + ri * xr).astype(np.float32)
    return ar + 1j * ai


def _mix_halves(X, R, xscale, rscale, flush_subnormals=False):
    def sp(v, s):
        h1, h2 = _split(v, s)
        if flush_subnormals:
            h1 = np.where(np.abs(h1) < 2.0 ** -14, 0.0, h1)
            h2 = np.where(np.abs(h2) < 2.0 ** -14, 0.0, h2)
        return h1, h2
    xr, xi, rr, ri = sp(X.real, xscale), sp(X.imag, xscale), sp(R.real, rscale), sp(R.imag, rscale)
    out = []
    for comp in (0, 1):
        lo = np.zeros(X.shape[::2], np.float64)
        hi = np.zeros_like(lo)
        for b in range(D):  # FP32 accumulation, the small products on their own (as the kernel's `lo` / `hi`)
            def term(i, j):
                if comp == 0:
                    return xr[i][:, b] * rr[j][b] - xi[i][:, b] * ri[j][b]
                return xi[i][:, b] * rr[j][b] + xr[i][:, b] * ri[j][b]
            lo = (lo + term(1, 0)).astype(np.float32).astype(np.float64)
            hi = (hi + term(0, 0)).astype(np.float32).astype(np.float64)
            lo = (lo + term(0, 1)).astype(np.float32).astype(np.float64)
        out.append(((hi + lo).astype(np.float32) / np.float32(xscale * rscale)).astype(np.float64))
    return out[0] + 1j * out[1]


@pytest.mark.parametrize("seed", [1, 2])
def test_two_half_split_mix_is_as_good_as_the_fp32_chain(seed):
    X, R, L = _operands(seed)
    Yex = np.einsum("sbm,bm->sm", X.astype(np.complex128), R.astype(np.complex128))
    rscale = 2.0 ** np.floor(np.log2(8192.0 / L))  # xl_poly_col_scale (xl_batch.cpp): the bound of the branch spectra under XLP_H_RMAX
    assert np.abs(R).max() * rscale <= 8192.0 and np.abs(X).max() * 128.0 < 65504.0
    e32 = _err(_mix_fp32_chain(X, R), Yex)
    eh = _err(_mix_halves(X, R, 128.0, rscale), Yex)
    ehf = _err(_mix_halves(X, R, 128.0, rscale, flush_subnormals=True), Yex)
    assert e32 < 5e-7 and eh < 3e-7 and ehf < 5e-7, (e32, eh, ehf)
    assert eh < 1.5 * e32, (eh, e32)


def test_unscaled_halves_would_not_do():
    """Why the scales exist: branch spectra of a unit-gain low-pass are ~1e-2 and their second halves fall into the half
    format's subnormal range; with them flushed the error is three orders of magnitude above the bar's margin."""
    X, R, _ = _operands(3, nseg=8)
    Yex = np.einsum("sbm,bm->sm", X.astype(np.complex128), R.astype(np.complex128))
    assert _err(_mix_halves(X, R, 1.0, 1.0, flush_subnormals=True), Yex) > 5e-5
