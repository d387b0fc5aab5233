"""CPU tests of the streaming-rule bookkeeping the kernels and the engine share (sdr-server_amd/csrc/xl_grid.h).

The header is plain C: it is compiled here with gcc behind a tiny shim and checked against brute force
(reference semantics: /root/reference/src/xlating.c:52-83 -- output k of a filter is produced in the call during
which sample k*D of ITS stream arrives; the phase is renormalised at the end of every call that produced output).
The second half models the control flow of the device-side NCO chain / consumer walk (xl_dev_inline.h:
xl_nco_client_chain, xl_phase_walk) in numpy float32 and checks it bit for bit against the oracle's phases over
multi-block calls -- the logic, not the HIP code itself (that is what the gpu tests do).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from pyoracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdr-server_amd", "csrc")
HCAP = 16384

SHIM = r"""
#include "xl_grid.h"
void g_dyn(uint32_t D, uint32_t T, uint32_t rem0, uint32_t hv0, uint32_t trel, uint32_t S, uint32_t G, uint32_t *out) {
  XlPos p = {trel, S, G, 0};
  XlDyn d = xl_grid_dyn(D, T, rem0, hv0, p);
  out[0] = d.base; out[1] = d.K; out[2] = d.zero_below; out[3] = d.j0;
}
uint32_t g_mstart(uint32_t j0, uint32_t D, uint32_t S, uint32_t g) { return xl_grid_mstart(j0, D, S, g); }
uint32_t g_bnd_next(uint32_t j0, uint32_t D, uint32_t S, uint32_t G, uint32_t K, uint32_t m) {
  XlBnd b = {j0, D, S, G, K};
  return xl_bnd_next(b, m);
}
uint32_t g_merge_j0(uint32_t j0_ref, uint32_t delta, uint32_t D) { return xl_merge_j0(j0_ref, delta, D); }
uint32_t g_merge_shift(uint32_t j0_ref, uint32_t delta, uint32_t D) { return xl_merge_shift(j0_ref, delta, D); }
uint32_t g_merge_points(uint32_t D, uint32_t S, uint32_t G) { XlPos p = {0, S, G, 0}; return xl_merge_points(D, p); }
"""


@pytest.fixture(scope="module")
def grid(tmp_path_factory):
    d = tmp_path_factory.mktemp("grid")
    src = d / "shim.c"
    src.write_text(SHIM)
    so = d / "libgrid.so"
    extra = os.environ.get("XL_SANITIZE_CFLAGS", "").split()  # (tools/sanitize.sh: -fsanitize=address,undefined)
    subprocess.run(["gcc", "-O1", "-Wall", "-Werror", "-shared", "-fPIC"] + extra + ["-I", CSRC, str(src), "-o", str(so)], check=True)
    L = C.CDLL(str(so))
    for n in ("g_mstart", "g_bnd_next", "g_merge_j0", "g_merge_shift", "g_merge_points"):
        getattr(L, n).restype = C.c_uint32
        getattr(L, n).argtypes = [C.c_uint32] * {"g_mstart": 4, "g_bnd_next": 6, "g_merge_j0": 3, "g_merge_shift": 3, "g_merge_points": 3}[n]
    L.g_dyn.argtypes = [C.c_uint32] * 7 + [C.POINTER(C.c_uint32)]
    return L


def brute_outputs(consumed, D, S, G):
    """Stream positions (client-local) of the outputs produced by a call of G blocks of S samples -> per-block lists."""
    per_block = [[] for _ in range(G)]
    first = -(-consumed // D) * D  # first multiple of D >= consumed
    n = first
    while n < consumed + G * S:
        per_block[(n - consumed) // S].append(n)
        n += D
    return per_block


def test_dyn_and_block_boundaries_match_brute_force(grid):
    rng = np.random.default_rng(7)
    out = (C.c_uint32 * 4)()
    for _ in range(400):
        D = int(rng.integers(1, 120))
        T = int(rng.integers(1, 600))
        G = int(rng.choice([1, 1, 2, 3, 8]))
        S = int(rng.integers(max(D, 1), 5000))
        consumed_at_plan = int(rng.integers(0, 100000))
        trel = int(rng.integers(0, 50)) * S * G
        consumed = consumed_at_plan + trel
        rem0, hv0 = consumed_at_plan % D, min(consumed_at_plan, HCAP)
        grid.g_dyn(D, T, rem0, hv0, trel, S, G, out)
        base, K, zero_below, j0 = out[0], out[1], out[2], out[3]
        blocks = brute_outputs(consumed, D, S, G)
        flat = [n for bl in blocks for n in bl]
        assert K == len(flat)
        assert j0 == (-consumed) % D
        assert zero_below == HCAP - min(consumed, HCAP)
        if flat:
            # window of output 0 in [history | blocks] coordinates: its newest sample is call-local j0
            assert base == HCAP + j0 - (T - 1)
            assert flat[0] - consumed == j0
        # block starts and the boundary function
        starts = np.cumsum([0] + [len(bl) for bl in blocks])
        for g in range(G + 1):
            assert grid.g_mstart(j0, D, S, g) == starts[g]
        for m in range(0, K, max(1, K // 37)):
            blk = int(np.searchsorted(starts, m, side="right")) - 1
            assert grid.g_bnd_next(j0, D, S, G, K, m) == starts[blk + 1], (m, blk)


def test_merged_grid_identity(grid):
    """Client output k == shared point q = k + shift evaluated with the taps delayed by delta (xl_grid.h)."""
    rng = np.random.default_rng(11)
    for _ in range(300):
        D = int(rng.integers(2, 90))
        T = int(rng.integers(2, 400))
        S = int(rng.integers(D, 4000))
        G = int(rng.choice([1, 2, 8]))
        j0_ref = int(rng.integers(0, D))
        delta = int(rng.integers(0, D))
        j0_c = grid.g_merge_j0(j0_ref, delta, D)
        shift = grid.g_merge_shift(j0_ref, delta, D)
        assert j0_c == (j0_ref + delta) % D
        base_ref = HCAP - (T - 1) + j0_ref - D     # first tap of shared point 0
        base_c = HCAP - (T - 1) + j0_c             # first tap of the client's output 0
        N = S * G
        K_c = -(-(N - j0_c) // D) if N > j0_c else 0
        Kq = grid.g_merge_points(D, S, G)
        for k in (0, 1, K_c - 1):
            if 0 <= k < K_c:
                q = k + shift
                assert base_ref + q * D + delta == base_c + k * D
                assert q < Kq
    # numerically: y_c[k] = sum_i r[i] x[base_c + k D + i] = sum_j r'[j] x[base_ref + q D + j], r'[j] = r[j - delta]
    D, T, delta, j0_ref = 7, 23, 5, 4
    r = rng.standard_normal(T) + 1j * rng.standard_normal(T)
    x = rng.standard_normal(HCAP + 2000) + 1j * rng.standard_normal(HCAP + 2000)
    rp = np.concatenate([np.zeros(delta), r])
    j0_c, shift = (j0_ref + delta) % D, (0 if j0_ref + delta >= D else 1)
    for k in range(20):
        a = np.dot(r, x[HCAP - (T - 1) + j0_c + k * D:][:T])
        q = k + shift
        b = np.dot(rp, x[HCAP - (T - 1) + j0_ref - D + q * D:][:T + delta])
        assert abs(a - b) < 1e-9


# ------------------------------------------------------------------------------------------------------------------
# Model of the device-side chain / walk control flow in float32


def f32(x):
    return np.float32(x)


def nco_next(p, inc):
    """xl_nco_next: t1 = p * inc.re, t2 = p * inc.im (each product rounded), p' = (t1.x - t2.y, t1.y + t2.x)"""
    t1x, t1y = f32(p[0] * inc[0]), f32(p[1] * inc[0])
    t2x, t2y = f32(p[0] * inc[1]), f32(p[1] * inc[1])
    return (f32(t1x - t2y), f32(t1y + t2x))


def renorm(p):
    mag = f32(np.sqrt(np.float64(p[0]) * np.float64(p[0]) + np.float64(p[1]) * np.float64(p[1])))
    return (f32(p[0] / mag), f32(p[1] / mag))


def bnd_next(j0, D, S, G, K, m):
    if G <= 1:
        return K
    g = (j0 + m * D) // S
    if g + 1 >= G:
        return K
    n = (g + 1) * S
    nb = -(-(n - j0) // D) if n > j0 else 0
    return min(nb, K)


def chain(p, inc, j0, D, S, G, K, kb, ke, tab):
    """xl_nco_client_chain: every 16th phase into tab (dict index -> phase), renormalised at block ends."""
    m = kb
    while m < ke:
        nb = bnd_next(j0, D, S, G, K, m)
        me = min(nb, ke)
        while m < me:
            if m % 16 == 0:
                tab[m // 16] = p
            p = nco_next(p, inc)
            m += 1
        if me == nb:
            p = renorm(p)
    return p


def walk(tab, inc, j0, D, S, G, K, m0, count):
    """xl_phase_walk: phases of outputs m0 .. m0 + count - 1 from the table entry below m0."""
    m = m0 & ~15
    p = tab[m // 16]
    nb = bnd_next(j0, D, S, G, K, m)
    out = []
    while m < m0 + count:
        if m >= m0:
            out.append(p)
        p = nco_next(p, inc)
        if m + 1 == nb:
            p = renorm(p)
            nb = bnd_next(j0, D, S, G, K, m + 1)
        m += 1
    return out


@pytest.mark.parametrize("D,G,S,slices", [(5, 3, 211, (0.0, 1.0)), (42, 4, 1303, (0.0, 0.12, 0.76, 1.0)), (7, 1, 500, (0.0, 0.5, 1.0))])
def test_chain_and_walk_reproduce_the_oracle_phases(D, G, S, slices):
    """G successive oracle calls == one chained call with block-end renormalisation; the consumers' walk from every
    16th phase reproduces each output's phase bit for bit.  The oracle's output for a constant-one input equals
    acc * phase with a constant acc once the history is full, so the phases are read from the oracle's own state."""
    code, taps = Oracle.lpf(1.0, 48000, 4800, 2000)
    assert code == 0
    o = Oracle(D, taps, -7000, 48000, 2 * S)
    inc = tuple(f32(v) for v in o.phase_incr)
    rng = np.random.default_rng(3)
    p = (f32(1.0), f32(0.0))
    consumed = 0
    for call in range(3):
        j0 = (-consumed) % D
        N = S * G
        K = -(-(N - j0) // D) if N > j0 else 0
        # phases the reference applies to each output of the G blocks, from G real oracle calls
        want = []
        q = p
        for g in range(G):
            x = rng.integers(0, 255, size=2 * S).astype(np.uint8)
            kg = len(o.process("cu8", x))
            for _ in range(kg):
                want.append(q)
                q = nco_next(q, inc)
            if kg:
                q = renorm(q)
            assert bits(q) == bits(o.phase), (call, g)  # the model's renormalised phase IS the oracle's state
        assert len(want) == K
        tab = {}
        state = p
        for a, b in zip(slices[:-1], slices[1:]):  # the call's chain in slices, as the three polyphase launches do
            kb = 0 if a == 0.0 else (int(K * a) & ~31)
            ke = K if b == 1.0 else (int(K * b) & ~31)
            state = chain(state, inc, j0, D, S, G, K, kb, ke, tab)
        assert bits(state) == bits(o.phase)
        for m0 in list(range(0, K, 13)) + [K - 1]:
            count = min(16, K - m0)
            got = walk(tab, inc, j0, D, S, G, K, m0, count)
            for i, ph in enumerate(got):
                assert bits(ph) == bits(want[m0 + i]), (call, m0, i)
        p = state
        consumed += N
    o.close()


def bits(p):
    return (np.float32(p[0]).tobytes(), np.float32(p[1]).tobytes())
