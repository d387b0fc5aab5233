"""CPU tests (-m "not gpu"): pin the oracle.

1. the CPU restatement (oracle/xlating_oracle.c) reproduces every golden vector the reference's own tests hold
   for this path, under the reference's own assertion semantics;
2. it is bit-identical to the committed outputs of the UNMODIFIED reference (tests/golden/live_*.npz) on all
   scenarios, including streaming state, ragged blocks, the even-tap quirk and the shared-history quirk;
3. when oracle/_ref is present (build container, and on the GPU box as a shipped .so) the same holds live.
"""
import numpy as np
import pytest

import scenarios
import siggen
import os

from conftest import GOLDEN, assert_ref_cf32, bits_equal, load_live, trunc1e4
from pyoracle import Oracle, RefLib


def olpf(*a):
    code, t = Oracle.lpf(*a)
    assert code == 0
    return t


def run_oracle(sc, sum_mode=0):
    taps = scenarios.make_taps(sc, lpf=olpf)
    o = Oracle(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"], sum_mode=sum_mode)
    outs = []
    for call in sc["calls"]:
        x = scenarios.to_path_input(sc, scenarios.make_input(sc, call))
        outs.append(o.process(sc["fmt"], x, call["out"]))
    return taps, outs, o


# ---------------------------------------------------------------- reference test arrays (G1-G7)


def test_lpf_golden_test_lpf_c(ref_vectors):
    """test/test_lpf.c:25-39"""
    exp = ref_vectors["test_lpf.c"]["test_lowpassTaps"]["expected_taps"]
    code, taps = Oracle.lpf(1.0, 8000, 1750, 500)
    assert code == 0 and taps.size == 39
    assert np.array_equal(trunc1e4(exp), trunc1e4(taps))


@pytest.mark.parametrize("args", [(0, 1750, 500), (8000, 5000, 500), (8000, 1750, 0)])
def test_lpf_bounds_test_lpf_c(args):
    """test/test_lpf.c:7-23: invalid arguments return -1"""
    code, taps = Oracle.lpf(1.0, *args)
    assert code == -1 and taps is None


def test_ntaps_table():
    """SURVEY A.6 tap-count table (lpf.c:31-38)"""
    L = Oracle.lib()
    assert L.orc_lpf_ntaps(2016000, 48000) == 101
    assert L.orc_lpf_ntaps(2016000, 9600) == 505
    assert L.orc_lpf_ntaps(2016000, 19200) == 253
    assert L.orc_lpf_ntaps(2016000, 2000) == 2429
    assert L.orc_lpf_ntaps(48000, 2000) == 57
    assert L.orc_lpf_ntaps(48000, 1920) == 61
    assert L.orc_lpf_ntaps(8000, 500) == 39


def test_g1_max_input_buffer_size(ref_vectors):
    """test/test_xlating.c:24-37"""
    v = ref_vectors["test_xlating.c"]["test_max_input_buffer_size"]
    _, outs, _ = run_oracle(scenarios.BY_NAME["g1_full"])
    assert_ref_cf32(v["expected_cf32"], outs[0])
    assert np.array_equal(np.asarray(v["expected_cs16"], np.int16), outs[1].reshape(-1))


def test_g2_partial_input_buffer_size(ref_vectors):
    """test/test_xlating.c:39-61"""
    v = ref_vectors["test_xlating.c"]["test_partial_input_buffer_size"]
    _, outs, _ = run_oracle(scenarios.BY_NAME["g2_partial"])
    assert_ref_cf32(v["expected_cf32"], outs[0])
    assert np.array_equal(np.asarray(v["expected_cs16"], np.int16), outs[1].reshape(-1))
    assert_ref_cf32(v["expected_next_cf32"], outs[2])
    assert np.array_equal(np.asarray(v["expected_next_cs16"], np.int16), outs[3].reshape(-1))


def test_g3_small_input_data():
    """test/test_xlating.c:63-81: one extra complex sample must not produce output"""
    _, outs, _ = run_oracle(scenarios.BY_NAME["g3_small"])
    assert len(outs[0]) == 20 and len(outs[1]) == 20
    assert len(outs[2]) == 0 and len(outs[3]) == 0


@pytest.mark.parametrize("name,fn", [("g4_rtl", "test_rtlsdr"), ("g5_airspy", "test_airspy"), ("g6_hackrf", "test_hackrf")])
def test_g4_g6_tcp_server_vectors(ref_vectors, name, fn):
    """test/test_tcp_server.c:154-248 end-to-end vectors, reproduced by calling lpf+xlating directly"""
    exp = ref_vectors["test_tcp_server.c"][fn]["expected"]
    _, outs, _ = run_oracle(scenarios.BY_NAME[name])
    assert_ref_cf32(exp, outs[0])


# ---------------------------------------------------------------- committed outputs of the unmodified reference


@pytest.mark.parametrize("sc", scenarios.SCENARIOS, ids=lambda s: s["name"])
def test_restatement_bit_exact_vs_committed_reference_outputs(sc):
    live = load_live(sc["name"])
    taps, outs, o = run_oracle(sc)
    assert bits_equal(taps, live["taps"])
    for ci, call in enumerate(sc["calls"]):
        assert len(outs[ci]) == int(live[f"n{ci}"]), (ci, len(outs[ci]), int(live[f"n{ci}"]))
        if call.get("keep", True):
            y = outs[ci] if call.get("keep_n") is None else outs[ci][: call["keep_n"]]
            assert bits_equal(y, live[f"y{ci}"]), f"{sc['name']} call {ci}"
    o.close()


@pytest.mark.parametrize("name", ["g9_default", "g11_t101", "g13_cf32_257"])
def test_f64_yardstick_close_to_canonical(name):
    """Accuracy yardstick: double-accumulated FIR vs canonical float32 order, max|d|/max|y| << 1e-5"""
    sc = scenarios.BY_NAME[name]
    _, a, _ = run_oracle(sc, sum_mode=0)
    _, b, _ = run_oracle(sc, sum_mode=1)
    for ci, call in enumerate(sc["calls"]):
        if call["out"] != "cf32" or len(a[ci]) == 0:
            continue
        err = np.abs(a[ci] - b[ci]).max() / np.abs(b[ci]).max()
        assert err < 2e-6, (name, ci, err)


def test_hypotf_double_form_matches_libm():
    """The NCO renormalisation kernel evaluates hypotf as (float)sqrt((double)x*x + (double)y*y); glibc's
    hypotf must agree on this box (xlating.c:73)."""
    import ctypes as C

    libm = C.CDLL("libm.so.6")
    libm.hypotf.argtypes = [C.c_float, C.c_float]
    libm.hypotf.restype = C.c_float
    L = Oracle.lib()
    rng = np.random.default_rng(7)
    th = rng.uniform(0, 2 * np.pi, 20000)
    r = 1 + rng.uniform(-1e-5, 1e-5, th.size)
    xs = (r * np.cos(th)).astype(np.float32)
    ys = (r * np.sin(th)).astype(np.float32)
    for x, y in zip(xs, ys):
        assert libm.hypotf(x, y) == L.orc_hypotf_via_double(x, y)


# ---------------------------------------------------------------- live reference (when oracle/_ref is present)


@pytest.mark.skipif(not RefLib.available("canon"), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("name", ["g1_full", "g3_small", "g9_default", "g12_even256"])
def test_restatement_bit_exact_vs_live_reference(name):
    sc = scenarios.BY_NAME[name]
    taps = scenarios.make_taps(sc, lpf=lambda *a: RefLib.lpf(*a)[1])
    ref = RefLib(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"])
    _, outs, _ = run_oracle(sc)
    for ci, call in enumerate(sc["calls"]):
        x = scenarios.make_input(sc, call)
        y = ref.process(sc.get("ref_fmt", sc["fmt"]), x, call["out"])
        assert bits_equal(y, outs[ci])


@pytest.mark.skipif(not RefLib.available("canon"), reason="oracle/_ref not built")
def test_lpf_bit_exact_vs_live_reference():
    for fs, cut, tw in [(2016000, 24000, 9600), (2016000, 24000, 48000), (2016000, 24000, 2000), (48000, 4800, 2000),
                        (8000, 1750, 500), (10000000, 50000, 20000), (2016000, 48000, 19200)]:
        a = Oracle.lpf(1.0, fs, cut, tw)[1]
        b = RefLib.lpf(1.0, fs, cut, tw)[1]
        assert bits_equal(a, b)


# ---- what "optimized" means on x86: the reference's AVX build (xlating.c:271-348) never renormalises the phase
# (:338-339), the canonical / NEON semantics (:73, :255) do every call.  The un-normalised float32 phasor drifts in
# MAGNITUDE by up to ~3e-8 per output, so the reference's own two builds agree within the 1e-5 bar only early in a
# stream, and how early depends on the increment's rounding: measured here on the unmodified builds, g9 shape
# (fc = -12 kHz) 1.2e-6 at block 9, 2.3e-5 at block 49, 1.5e-3 at block 399 (tests/golden/fast_divergence.json);
# g10 shape (6242 outputs per block) 1.0e-5 already at block 3; g11 shape (fc = -500 kHz) 6.8e-5 at block 1.
# Nothing can stay within 1e-5 of both builds.  This library follows the renormalising semantics; the fast fixtures
# pin it to the AVX build (a) outright on block 0, before any renormalisation has happened, and on all ten blocks of the
# g9 shape, and (b) on every block of every shape up to a complex scale factor per 64 outputs (the drifted phasor).
def fast_fixture_errors(name, process):
    """-> per block (raw max|d|/max|y|, the same after removing a least-squares complex scale per 64 outputs, last |scale|)"""
    sc = scenarios.BY_NAME[name]
    fx = np.load(os.path.join(GOLDEN, f"fast_{name}.npz"))
    res = []
    for k in range(scenarios.FAST_BLOCKS):
        y = process(scenarios.fast_block(sc, k)).astype(np.complex128)
        assert len(y) == int(fx[f"n{k}"])
        if k == scenarios.FAST_BLOCKS - 1:
            a, b = y, fx[f"y{k}"].astype(np.complex128)
        else:
            a = np.concatenate([y[:scenarios.FAST_HEAD], y[-scenarios.FAST_TAIL:]])
            b = np.concatenate([fx[f"head{k}"], fx[f"tail{k}"]]).astype(np.complex128)
        scale = np.abs(y).max()
        resid, r = 0.0, 1.0
        for c in range(0, len(a), 64):  # the AVX phasor drifts up to ~3e-8 per output: one scale per 64 outputs
            aa, bb = a[c:c + 64], b[c:c + 64]
            r = np.vdot(bb, aa) / np.vdot(bb, bb)
            resid = max(resid, float(np.abs(aa - r * bb).max() / scale))
        res.append((float(np.abs(a - b).max() / scale), resid, float(abs(r))))
    return res


def check_fast_fixture(name, process, tol=1e-5):
    res = fast_fixture_errors(name, process)
    assert res[0][0] <= tol, (name, res[0])                       # block 0: same semantics, outright
    if name == "g9_default":
        assert max(r[0] for r in res) <= tol, (name, res)         # fc = -12 kHz: the AVX phasor barely drifts in 10 blocks
    assert max(r[1] for r in res) <= tol, (name, res)             # every block, up to the drifted phasor's scale
    return res


@pytest.mark.parametrize("name", scenarios.FAST_SHAPES)
def test_oracle_vs_the_reference_avx_build(name):
    sc = scenarios.BY_NAME[name]
    fx = np.load(os.path.join(GOLDEN, f"fast_{name}.npz"))
    o = Oracle(sc["D"], fx["taps"], sc["fc"], sc["fs"], sc["max_input"])
    res = check_fast_fixture(name, lambda x: o.process(sc["fmt"], x))
    o.close()
    if name == "g11_t101":  # the documented drift: the AVX build leaves its own 1e-5 neighbourhood after one block
        assert res[1][0] > 1e-5 and abs(res[9][2] - 1.0) > 1e-4, res


def _x86_block_error(y, fx, k, head, tail):
    if f"y{k}" in fx:
        pairs = [(y, fx[f"y{k}"])]
    else:
        pairs = [(y[:head], fx[f"head{k}"]), (y[-tail:], fx[f"tail{k}"])]
    scale = max(np.abs(b).max() for _, b in pairs)
    return max(float(np.abs(a.astype(np.complex128) - b).max() / scale) for a, b in pairs)


def check_x86_fixtures(make_process, flavour, tol=1e-5):
    """The x86 AVX build's process_optimized_* (phase never renormalised, xlating.c:338-339) pinned OUTRIGHT -- max|d| /
    max|y| <= tol per block, no scale factor -- to outputs of the UNMODIFIED reference:
      flavour "fma":   the build with FMA enabled (oracle/_ref/libref_fast.so: -O3 -ffast-math -mavx2 -mfma): all ten blocks of
                       the three fast_*.npz shapes and the sampled blocks of the 400-block stream x86_long_g9.npz
      flavour "plain": the build without FMA (libref_avx.so: the reference's Release flags + -mavx): all ten blocks of
                       avx_g11_t101.npz -- the shape on which the two builds' phase steps part ways within a few blocks.
    make_process(sc, taps) -> (process(x) -> complex64[K], close())."""
    worst = 0.0
    files = [(n, f"fast_{n}.npz") for n in scenarios.FAST_SHAPES] if flavour == "fma" else [("g11_t101", "avx_g11_t101.npz")]
    for name, fname in files:
        sc = scenarios.BY_NAME[name]
        fx = np.load(os.path.join(GOLDEN, fname))
        process, close = make_process(sc, fx["taps"])
        for k in range(scenarios.FAST_BLOCKS):
            y = process(scenarios.fast_block(sc, k))
            assert len(y) == int(fx[f"n{k}"]), (name, k)
            err = _x86_block_error(y, fx, k, scenarios.FAST_HEAD, scenarios.FAST_TAIL)
            assert err <= tol, (fname, k, err)
            worst = max(worst, err)
        close()
    if flavour != "fma":
        return worst
    sc = scenarios.BY_NAME["g9_default"]
    fx = np.load(os.path.join(GOLDEN, "x86_long_g9.npz"))
    process, close = make_process(sc, fx["taps"])
    for k in range(scenarios.X86_LONG_BLOCKS):
        y = process(scenarios.fast_block(sc, k % 16))
        if k not in scenarios.X86_LONG_SAMPLED:
            continue
        assert len(y) == int(fx[f"n{k}"]), k
        err = _x86_block_error(y, fx, k, scenarios.X86_HEAD, scenarios.X86_TAIL)
        assert err <= tol, ("x86_long", k, err)
        worst = max(worst, err)
    close()
    return worst


@pytest.mark.parametrize("flavour", ["fma", "plain"])
def test_oracle_without_renormalisation_is_the_x86_build(flavour):
    """orc_xlating_set_renorm(f, 0) (+ set_fma_step for the FMA build) == the reference's AVX process_optimized_* over short
    AND long streams, outright -- and the WRONG step does not pass (the two builds really differ)."""

    def make(fma):
        def mk(sc, taps):
            o = Oracle(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"], renorm=False, fma_step=fma)
            return (lambda x: o.process(sc["fmt"], x)), o.close
        return mk

    assert check_x86_fixtures(make(flavour == "fma"), flavour) <= 2e-6
    with pytest.raises(AssertionError):
        check_x86_fixtures(make(flavour != "fma"), flavour)


def test_renormalising_semantics_leave_the_x86_stream():
    """WITH the per-call renormalisation the same stream is 1.5e-3 away from the x86 build's by block 399 (why the mode exists)."""
    sc = scenarios.BY_NAME["g9_default"]
    fx = np.load(os.path.join(GOLDEN, "x86_long_g9.npz"))
    o = Oracle(sc["D"], fx["taps"], sc["fc"], sc["fs"], sc["max_input"])
    for k in range(scenarios.X86_LONG_BLOCKS):
        y = o.process(sc["fmt"], scenarios.fast_block(sc, k % 16))
    o.close()
    assert float(np.abs(y - fx["y399"]).max() / np.abs(fx["y399"]).max()) > 1e-3


@pytest.mark.skipif(not (RefLib.available("fast") and RefLib.available("avx")), reason="oracle/_ref not built")
def test_phase_step_of_both_reference_builds_bit_for_bit():
    """How the two steps were established.  A one-tap filter at decimation 1 fed a constant 0.5 + 0j returns 0.5 * phase[k]
    exactly, so the unmodified reference hands out its phase sequence: the build without FMA follows the plain product, the
    build with FMA follows re = fma(pr, ir, -(pi * ii)), im = fma(pr, ii, pi * ir) -- 4000 steps, bit for bit."""
    taps = np.array([1.0], np.float32)
    fs, fc, n = 2016000, -500000, 4000
    x = np.zeros(2 * n, np.int16)
    x[0::2] = 16384
    for flav, fma in (("avx", False), ("fast", True)):
        r = RefLib(1, taps, fc, fs, 4 * n, flavour=flav, variant="optimized")
        want = r.process("cs16", x, "cf32")
        r.close()
        o = Oracle(1, taps, fc, fs, 4 * n, renorm=False, fma_step=fma)
        got = o.process("cs16", x)
        o.close()
        assert bits_equal(got, want), flav
        o = Oracle(1, taps, fc, fs, 4 * n, renorm=False, fma_step=not fma)
        assert not bits_equal(o.process("cs16", x), want), flav
        o.close()


def test_reference_builds_diverge_beyond_tolerance_later():
    import json

    d = json.load(open(os.path.join(GOLDEN, "fast_divergence.json")))["block"]
    assert d["9"] < 1e-5 < d["49"] < d["399"]  # the documented reason why only the first blocks are pinned to the AVX build


def test_population_driver_equals_single_filters():
    """oracle/population.c (threaded, one filter per client) == the same filters run one by one: warm blocks, compared
    blocks, the fast-forward, two input formats."""
    from pyoracle import population

    code, taps = Oracle.lpf(1.0, 2016000, 24000, 9600)
    fcs = [-984000 + 1920 * c for c in range(19)]
    x = siggen.xs_u8(5, 3 * 60002)
    res = population(42, taps, fcs, 2016000, 60002, "cu8", x, 2, nwarm=1, skip_fresh=30001, skip_calls=5, threads=3)
    for c in (0, 7, 18):
        o = Oracle(42, taps, fcs[c], 2016000, 60002)
        o.skip_calls(30001, 5)
        bl = np.split(x, 3)
        o.process("cu8", bl[0])
        want = np.concatenate([o.process("cu8", bl[1]), o.process("cu8", bl[2])])
        assert bits_equal(res[c], want), c
        o.close()
    y = siggen.xs_s16(6, 2 * 4000)
    res = population(42, taps, fcs[:3], 2016000, 8000, "cs16", y, 2)
    o = Oracle(42, taps, fcs[2], 2016000, 8000)
    want = np.concatenate([o.process("cs16", b) for b in np.split(y, 2)])
    assert bits_equal(res[2], want)
    o.close()
