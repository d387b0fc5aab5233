"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI of include/xlating.h, against
  (1) the oracle run live on the same inputs      -- native: BIT-EXACT;  optimized: max|d|/max|y| <= 1e-5
  (2) the committed outputs of the unmodified reference (tests/golden/live_*.npz) -- same bars
  (3) the expected arrays of the reference's own tests under the reference's own assertion (G1-G6).
cs16 (Q15) outputs are exact integers: bit-exact in every variant.
Tolerance: BASELINE.json north_star "within 1e-5 relative", read as max|d| / max|y_ref| per call.
"""
import numpy as np
import pytest

import scenarios
import sdr_server_amd as xl
import os

from conftest import GOLDEN, assert_ref_cf32, bits_equal, load_live
from pyoracle import Oracle

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


def hip_lpf(*a):
    code, t = xl.create_low_pass_filter(*a)
    assert code == 0
    return t


def run_hip(sc, variant):
    taps = scenarios.make_taps(sc, lpf=hip_lpf)
    f = xl.XlatingFilter(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"])
    outs = []
    for call in sc["calls"]:
        x = scenarios.to_path_input(sc, scenarios.make_input(sc, call))
        outs.append(f.process(variant, sc["fmt"], call["out"], x))
    f.close()
    return taps, outs


def run_oracle(sc):
    code_taps = scenarios.make_taps(sc, lpf=lambda *a: Oracle.lpf(*a)[1])
    o = Oracle(sc["D"], code_taps, sc["fc"], sc["fs"], sc["max_input"])
    outs = []
    for call in sc["calls"]:
        x = scenarios.to_path_input(sc, scenarios.make_input(sc, call))
        outs.append(o.process(sc["fmt"], x, call["out"]))
    o.close()
    return outs


def rel_err(a, b):
    if len(b) == 0:
        return 0.0
    return float(np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max() / max(np.abs(b).max(), 1e-30))


FAST = [s for s in scenarios.SCENARIOS if s["name"] != "g15_drift"]


def test_device_is_gfx950():
    assert xl.simd_status() == "HIP gfx950"
    assert "gfx950" in xl.device_info(), xl.device_info()


@pytest.mark.parametrize("sc", FAST, ids=lambda s: s["name"])
def test_native_bit_exact_vs_oracle_and_reference_fixture(sc):
    taps, outs = run_hip(sc, "native")
    want = run_oracle(sc)
    live = load_live(sc["name"])
    assert bits_equal(taps, live["taps"])
    for ci, call in enumerate(sc["calls"]):
        assert len(outs[ci]) == int(live[f"n{ci}"]), (sc["name"], ci)
        assert bits_equal(outs[ci], want[ci]), f"{sc['name']} call {ci} ({call['out']}) differs from oracle"
        if call.get("keep", True):
            y = outs[ci] if call.get("keep_n") is None else outs[ci][: call["keep_n"]]
            assert bits_equal(y, live[f"y{ci}"]), f"{sc['name']} call {ci} differs from committed reference output"


@pytest.mark.parametrize("sc", FAST, ids=lambda s: s["name"])
def test_optimized_within_tolerance(sc):
    _, outs = run_hip(sc, "optimized")
    want = run_oracle(sc)
    for ci, call in enumerate(sc["calls"]):
        assert len(outs[ci]) == len(want[ci])
        if call["out"] == "cs16":
            assert bits_equal(outs[ci], want[ci])  # Q15: optimized == native == exact (xlating.c:437-447)
        else:
            assert rel_err(outs[ci], want[ci]) <= REL_TOL, (sc["name"], ci, rel_err(outs[ci], want[ci]))


@pytest.mark.parametrize("variant", ["native", "optimized"])
def test_reference_test_arrays_g1_g6(ref_vectors, variant):
    """The reference's own assertions (test/utils.c:176-189) on test/test_xlating.c + test/test_tcp_server.c arrays."""
    v = ref_vectors["test_xlating.c"]
    _, o = run_hip(scenarios.BY_NAME["g1_full"], variant)
    assert_ref_cf32(v["test_max_input_buffer_size"]["expected_cf32"], o[0])
    assert np.array_equal(np.asarray(v["test_max_input_buffer_size"]["expected_cs16"], np.int16), o[1].reshape(-1))
    _, o = run_hip(scenarios.BY_NAME["g2_partial"], variant)
    p = v["test_partial_input_buffer_size"]
    assert_ref_cf32(p["expected_cf32"], o[0])
    assert np.array_equal(np.asarray(p["expected_cs16"], np.int16), o[1].reshape(-1))
    assert_ref_cf32(p["expected_next_cf32"], o[2])
    assert np.array_equal(np.asarray(p["expected_next_cs16"], np.int16), o[3].reshape(-1))
    _, o = run_hip(scenarios.BY_NAME["g3_small"], variant)
    assert len(o[0]) == 20 and len(o[1]) == 20 and len(o[2]) == 0 and len(o[3]) == 0
    for name, fn in (("g4_rtl", "test_rtlsdr"), ("g5_airspy", "test_airspy"), ("g6_hackrf", "test_hackrf")):
        _, o = run_hip(scenarios.BY_NAME[name], variant)
        assert_ref_cf32(ref_vectors["test_tcp_server.c"][fn]["expected"], o[0])


def test_nco_drift_100_blocks_native_bit_exact():
    """G15: 100 server-default blocks on one filter; the float32 phase recurrence + per-call renormalisation must
    track the reference bit for bit (an 'ideal' NCO would be 5e-5 off after one block)."""
    sc = scenarios.BY_NAME["g15_drift"]
    taps, outs = run_hip(sc, "native")
    live = load_live(sc["name"])
    for ci, call in enumerate(sc["calls"]):
        assert len(outs[ci]) == int(live[f"n{ci}"])
        if call.get("keep", True):
            y = outs[ci] if call.get("keep_n") is None else outs[ci][: call["keep_n"]]
            assert bits_equal(y, live[f"y{ci}"]), f"block {ci}"


def test_nco_drift_100_blocks_optimized_within_tolerance():
    sc = scenarios.BY_NAME["g15_drift"]
    _, outs = run_hip(sc, "optimized")
    live = load_live(sc["name"])
    worst = 0.0
    for ci, call in enumerate(sc["calls"]):
        if call.get("keep", True):
            y = outs[ci] if call.get("keep_n") is None else outs[ci][: call["keep_n"]]
            ref = live[f"y{ci}"]
            worst = max(worst, float(np.abs(y - ref).max() / np.abs(live["y99"]).max()))
    assert worst <= REL_TOL, worst


def test_optimized_accuracy_vs_f64_yardstick():
    """The fused variant should be at least as close to a double-accumulated FIR as the canonical float32 order."""
    sc = scenarios.BY_NAME["g9_default"]
    taps = scenarios.make_taps(sc, lpf=hip_lpf)
    x = scenarios.make_input(sc, sc["calls"][0])
    o64 = Oracle(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"], sum_mode=1)
    o32 = Oracle(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"], sum_mode=0)
    y64 = o64.process("cu8", x)
    y32 = o32.process("cu8", x)
    f = xl.XlatingFilter(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"])
    yo = f.process("optimized", "cu8", "cf32", x)
    f.close()
    e_fast, e_canon = rel_err(yo, y64), rel_err(y32, y64)
    assert e_fast <= 2e-6 and e_fast <= 1.5 * e_canon + 1e-7, (e_fast, e_canon)


@pytest.mark.parametrize("fmt", ["cu8", "cs8", "cs16"])
def test_empty_and_tiny_inputs(fmt):
    """input_len 0, 1 element, 1 sample: no output, state intact, later output still bit-exact."""
    taps = hip_lpf(1.0, 48000, 4800, 2000)
    f = xl.XlatingFilter(5, taps, -12000, 48000, 4000)
    o = Oracle(5, taps, -12000, 48000, 4000)
    gen = {"cu8": scenarios.siggen.ramp_u8, "cs8": scenarios.siggen.ramp_s8, "cs16": scenarios.siggen.ramp_s16}[fmt]
    for n in (0, 1, 2, 0, 3, 400, 2, 2, 2, 2, 2, 2000):
        x = gen(n * 3, n)
        a, b = f.process("native", fmt, "cf32", x), o.process(fmt, x, "cf32")
        assert bits_equal(a, b), n
        a, b = f.process("native", fmt, "cs16", x), o.process(fmt, x, "cs16")
        assert bits_equal(a, b), n
    f.close()
    o.close()


def test_maximum_block_server_default():
    """Largest block the server feeds (buffer_size 262144 bytes, config.conf:13) for all three device formats."""
    taps = hip_lpf(1.0, 2016000, 24000, 9600)
    for fmt, x in (("cu8", scenarios.siggen.xs_u8(11, 262144)), ("cs8", scenarios.siggen.xs_s8(12, 262144)),
                   ("cs16", scenarios.siggen.xs_s16(13, 131072))):
        f = xl.XlatingFilter(42, taps, 123456, 2016000, 262144)
        o = Oracle(42, taps, 123456, 2016000, 262144)
        for _ in range(2):
            assert bits_equal(f.process("native", fmt, "cf32", x), o.process(fmt, x, "cf32"))
        f.close()
        o.close()


def test_many_filters_threads():
    """dsp_worker model: N threads, each with its own filter, concurrently (SURVEY 8(b) threading)."""
    import threading

    taps = hip_lpf(1.0, 2016000, 24000, 9600)
    x = scenarios.siggen.xs_u8(5, 262144)
    o = Oracle(42, taps, -12000, 2016000, 262144)
    want = [o.process("cu8", x) for _ in range(3)]
    o.close()
    errs = []

    def worker():
        try:
            f = xl.XlatingFilter(42, taps, -12000, 2016000, 262144)
            for k in range(3):
                y = f.process("native", "cu8", "cf32", x)
                if not bits_equal(y, want[k]):
                    errs.append(k)
            f.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=worker) for _ in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


@pytest.mark.parametrize("variant", ["native", "optimized"])
def test_huge_decimation_window_exceeds_lds(variant):
    """HackRF-style 20 Msps -> 50 kHz: D = 400, 4819 taps.  A 64-output window image would need 240 KB of LDS, so the
    launch falls back to 32 outputs per wave (upper lanes idle).  Single filter and batch engine vs the oracle."""
    fs, rate = 20000000, 50000
    taps = hip_lpf(1.0, fs, rate // 2, rate // 5)
    assert taps.size == 4819
    D = fs // rate
    nbytes = 131072
    f = xl.XlatingFilter(D, taps, -3000000, fs, nbytes)
    o = Oracle(D, taps, -3000000, fs, nbytes)
    eng = xl.BatchEngine(fs, "cs8", nbytes)
    ob = {}
    for c in range(3):
        ob[eng.add_client(D, taps, 1000000 * c - 500000)] = Oracle(D, taps, 1000000 * c - 500000, fs, nbytes)
    for k in range(3):
        x = scenarios.siggen.xs_s8(600 + k, nbytes if k != 1 else 50002)
        got, want = f.process(variant, "cs8", "cf32", x), o.process("cs8", x)
        eng.process_host(x, variant)
        eng.fetch()
        if variant == "native":
            assert bits_equal(got, want), k
            for cid, oc in ob.items():
                assert bits_equal(eng.output(cid), oc.process("cs8", x)), (k, cid)
        else:
            assert len(got) == len(want) and rel_err(got, want) <= REL_TOL
            for cid, oc in ob.items():
                w = oc.process("cs8", x)
                assert rel_err(eng.output(cid), w) <= REL_TOL, (k, cid)
    f.close()
    o.close()
    eng.close()


def test_rejects_shapes_that_cannot_fit():
    """T - 1 beyond the engine's raw-history capacity is refused with -EINVAL (not silently wrong)."""
    taps = np.ones(20000, np.float32) / 20000
    eng = xl.BatchEngine(2016000, "cu8", 262144)
    with pytest.raises(xl.XlatingError) as e:
        eng.add_client(42, taps, 0)
    assert e.value.code == -22
    eng.close()


@pytest.mark.parametrize("seed", list(range(1, 1 + int(__import__("os").environ.get("XL_TEST_FUZZ_SEEDS", "6")))))
def test_randomised_dropin_filter_vs_oracle(seed):
    """Randomised call sequences on ONE drop-in filter: any input format, cf32 and cs16 outputs interleaved (they share
    the history and the NCO phase, xlating.c:84-89), native and optimized calls in any order, block lengths from nothing
    to the maximum, odd and even tap counts.  Every call against the oracle: bit-exact for native cf32 and for every
    cs16 output, 1e-5 relative for optimized cf32."""
    import siggen
    rng = np.random.default_rng(7000 + seed)
    fmt = ["cu8", "cs8", "cs16", "cf32"][int(rng.integers(4))]
    D, taps = [(42, hip_lpf(1.0, 2016000, 24000, 9600)), (21, hip_lpf(1.0, 2016000, 48000, 19200)),
               (7, siggen.hamming_sinc(64, 0.05)), (5, siggen.hamming_sinc(57, 0.08)), (100, siggen.hamming_sinc(301, 0.004)),
               (1, siggen.hamming_sinc(33, 0.2))][int(rng.integers(6))]
    fs = 2016000
    fc = int(rng.integers(-900000, 900000))
    max_input = int(rng.choice([4096, 65536, 262144]))
    f = xl.XlatingFilter(D, taps, fc, fs, max_input)
    o = Oracle(D, taps, fc, fs, max_input)
    per = {"cu8": 1, "cs8": 1, "cs16": 1, "cf32": 1}[fmt]  # max_input counts elements of the input type
    worst = 0.0
    for k in range(int(rng.integers(6, 16))):
        r = rng.random()
        n = max_input if r < 0.2 else int(rng.integers(0, 40)) if r < 0.35 else int(rng.integers(0, max_input // per + 1))
        n -= n % 2  # whole IQ pairs
        if fmt == "cu8":
            x = siggen.xs_u8(seed * 100 + k, n)
        elif fmt == "cs8":
            x = siggen.xs_u8(seed * 100 + k, n).view(np.int8)
        elif fmt == "cs16":
            x = siggen.xs_s16(seed * 100 + k, n)
        else:
            x = (siggen.xs_s16(seed * 100 + k, n).astype(np.float32) / np.float32(32768)).astype(np.float32)
        out = "cs16" if (rng.random() < 0.3 and fmt != "cf32") else "cf32"
        variant = "native" if rng.random() < 0.4 else "optimized"
        got = f.process(variant, fmt, out, x)
        want = o.process(fmt, x, out)
        assert len(got) == len(want), (k, fmt, out, n)
        if out == "cs16" or variant == "native":
            assert bits_equal(got, want), (k, fmt, out, variant, n)
        else:
            worst = max(worst, rel_err(got, want))
    assert worst <= REL_TOL, worst
    f.close()
    o.close()


# ---- "optimized" pinned against the reference's x86 AVX build (tests/golden/fast_*.npz, see tests/test_oracle.py):
# outright on block 0 and on all ten blocks of the g9 shape; on every block up to the drifted phasor's scale.
@pytest.mark.parametrize("name", scenarios.FAST_SHAPES)
@pytest.mark.parametrize("path", ["dropin", "batch_direct", "batch_polyphase"])
def test_optimized_vs_reference_avx_build_fixtures(name, path):
    from test_oracle import check_fast_fixture

    sc = scenarios.BY_NAME[name]
    fx = np.load(os.path.join(GOLDEN, f"fast_{name}.npz"))
    if path == "dropin":
        f = xl.XlatingFilter(sc["D"], fx["taps"], sc["fc"], sc["fs"], sc["max_input"])
        check_fast_fixture(name, lambda x: f.process("optimized", sc["fmt"], "cf32", x))
        f.close()
        return
    eng = xl.BatchEngine(sc["fs"], sc["fmt"], sc["max_input"])
    eng.set_option("polyphase", 1 if path == "batch_polyphase" else 0)
    cid = [eng.add_client(sc["D"], fx["taps"], sc["fc"]) for _ in range(3)][1]

    def process(x):
        eng.process_host(x, "optimized")
        eng.fetch()
        return eng.output(cid)

    check_fast_fixture(name, process)
    if path == "batch_polyphase":
        assert "polyphase: cls0" in eng.describe(), eng.describe()
    eng.close()


# ---- the x86 build's semantics as a selectable mode: never renormalise (xlating.c:338-339), pinned OUTRIGHT to the reference's
# AVX build: all ten blocks of the three fast_*.npz shapes and the sampled blocks of a 400-block stream (x86_long_g9.npz)
@pytest.mark.parametrize("flavour", ["fma", "plain"])
@pytest.mark.parametrize("path", ["dropin", "batch_direct", "batch_polyphase"])
def test_optimized_x86_mode_vs_reference_avx_build_long_stream(path, flavour):
    from test_oracle import check_x86_fixtures

    mode = "optimized_x86_fma" if flavour == "fma" else "optimized_x86"

    def make(sc, taps):
        if path == "dropin":
            f = xl.XlatingFilter(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"])
            f.set_optimized_x86(2 if flavour == "fma" else 1)
            return (lambda x: f.process("optimized", sc["fmt"], "cf32", x)), f.close
        eng = xl.BatchEngine(sc["fs"], sc["fmt"], sc["max_input"])
        eng.set_option("polyphase", 0 if path == "batch_direct" else 1)
        cid = [eng.add_client(sc["D"], taps, sc["fc"]) for _ in range(3)][1]

        def process(x):
            eng.process_host(x, mode)
            eng.fetch()
            return eng.output(cid)

        return process, eng.close

    assert check_x86_fixtures(make, flavour) <= 1e-5


@pytest.mark.parametrize("flavour", ["plain", "fma"])
def test_optimized_x86_mode_groups_of_blocks(flavour):
    """XL_MODE_OPTIMIZED_X86 / _X86_FMA in calls of four blocks (polyphase launches + the side-stream chain kernel, which must not
    renormalise at the block ends inside a call either), a class of 160 clients; sampled clients vs the oracle with the
    renormalisation switched off, over 24 blocks; then a renormalising call in between keeps working (shared phase)."""
    from pyoracle import Oracle
    import siggen

    code, t48 = xl.create_low_pass_filter(1.0, 2016000, 24000, 9600)
    n, G = 100002, 4
    eng = xl.BatchEngine(2016000, "cu8", n, group_blocks=G)
    fcs = [-900000 + 11000 * c for c in range(160)]
    ids = [eng.add_client(42, t48, fc) for fc in fcs]
    sample = [0, 63, 64, 159]
    ors = {c: Oracle(42, t48, fcs[c], 2016000, n, renorm=False, fma_step=flavour == "fma") for c in sample}
    for k in range(6):
        x = siggen.xs_u8(9500 + k, G * n)
        eng.process_host_group(x, G, "optimized_x86_fma" if flavour == "fma" else "optimized_x86")
        eng.fetch()
        for c, o in ors.items():
            want = np.concatenate([o.process("cu8", bl) for bl in np.split(x, G)])
            got = eng.output(ids[c])
            assert len(got) == len(want) and rel_err(got, want) <= REL_TOL, (k, c, rel_err(got, want))
    assert "polyphase: cls0 D42 T505 cols160" in eng.describe(), eng.describe()
    # the phase drifted away from 1 in amplitude by now and both engines agree on it
    for c, o in ors.items():
        got, want = eng.phase(ids[c]), o.phase
        assert tuple(np.float32(v).tobytes() for v in got) == tuple(np.float32(v).tobytes() for v in want), c
    eng.close()
