"""Parity scenarios for the xlating hot path.

One table drives three consumers:
  * tests/golden/make_golden.py  -- runs the UNMODIFIED reference on each scenario, commits outputs
  * tests/test_oracle.py         -- CPU restatement vs those outputs (bit-exact) and vs the reference tests' arrays
  * tests/test_gpu_parity.py     -- HIP path (through the C-ABI) vs oracle / fixtures

A scenario is a filter configuration plus an ordered list of process() calls on ONE filter instance, so
streaming state (history, NCO phase, the shared history counter of the two output families) is exercised.
"""
import numpy as np

import siggen

SEED = siggen.XS_SEED


def _calls(*specs):
    out = []
    for s in specs:
        gen, a, n, o = s[:4]
        d = {"gen": gen, "a": a, "n": n, "out": o}
        if len(s) > 4:
            d.update(s[4])
        out.append(d)
    return out


SCENARIOS = [
    # ---- the reference's own unit-test shapes (test/test_xlating.c:15-81) ---------------------------------
    dict(name="g1_full", fs=48000, D=5, fc=-12000, max_input=2000, fmt="cu8", taps=("lpf", 4800, 2000),
         calls=_calls(("ramp", 0, 2000, "cf32"), ("ramp", 0, 2000, "cs16"))),
    dict(name="g2_partial", fs=48000, D=5, fc=-12000, max_input=2000, fmt="cu8", taps=("lpf", 4800, 2000),
         calls=_calls(("ramp", 0, 200, "cf32"), ("ramp", 0, 200, "cs16"), ("ramp", 200, 200, "cf32"), ("ramp", 200, 200, "cs16"))),
    dict(name="g3_small", fs=48000, D=5, fc=-12000, max_input=2000, fmt="cu8", taps=("lpf", 4800, 2000),
         calls=_calls(("ramp", 0, 198, "cf32"), ("ramp", 0, 198, "cs16"), ("ramp", 200, 2, "cf32"), ("ramp", 200, 2, "cs16"),
                      # beyond the reference test: keep going so the shared-history quirk shows in data
                      ("ramp", 7, 300, "cf32"), ("ramp", 9, 300, "cs16"), ("ramp", 11, 46, "cf32"))),
    # ---- end-to-end vectors of test/test_tcp_server.c:154-248 (61 taps, D=5) ---------------------------------
    dict(name="g4_rtl", fs=48000, D=5, fc=-12000, max_input=131072, fmt="cu8", taps=("lpf", 4800, 1920),
         calls=_calls(("ramp", 0, 200, "cf32"))),
    dict(name="g5_airspy", fs=48000, D=5, fc=-12000, max_input=131072, fmt="cs16", taps=("lpf", 4800, 1920),
         calls=_calls(("ramp", 0, 200, "cf32"))),
    dict(name="g6_hackrf", fs=48000, D=5, fc=-12000, max_input=131072, fmt="cs8", taps=("lpf", 4800, 1920),
         calls=_calls(("ramp", 0, 200, "cf32"))),
    # ---- test/perf_xlating.c shape: 2429 taps, D=42, 200000-byte staircase -------------------------------------
    dict(name="g8_perf", fs=2016000, D=42, fc=-12000, max_input=200000, fmt="cu8", taps=("lpf", 24000, 2000),
         calls=_calls(("stair", 0, 200000, "cf32"), ("stair", 0, 200000, "cf32"), ("stair", 0, 200000, "cf32"),
                      ("stair", 0, 200000, "cs16"))),
    # ---- server default: 505 taps, D=42, 262144-byte blocks, one ragged block ----------------------------------
    dict(name="g9_default", fs=2016000, D=42, fc=-12000, max_input=262144, fmt="cu8", taps=("lpf", 24000, 9600),
         calls=_calls(("xs", SEED, 262144, "cf32"), ("xs", SEED + 1, 100002, "cf32"), ("xs", SEED + 2, 262144, "cf32"),
                      ("xs", SEED + 3, 262144, "cs16"), ("xs", SEED + 4, 2, "cf32"), ("xs", SEED + 5, 262144, "cf32"))),
    dict(name="g10_96k", fs=2016000, D=21, fc=345678, max_input=262144, fmt="cu8", taps=("lpf", 48000, 19200),
         calls=_calls(("xs", SEED + 10, 262144, "cf32"), ("xs", SEED + 11, 262144, "cf32"))),
    dict(name="g11_t101", fs=2016000, D=42, fc=-500000, max_input=262144, fmt="cs16", taps=("lpf", 24000, 48000),
         calls=_calls(("xs", SEED + 20, 131072, "cf32"), ("xs", SEED + 21, 131072, "cf32"), ("xs", SEED + 22, 131072, "cs16"))),
    dict(name="g16_cs8", fs=2016000, D=42, fc=700001, max_input=262144, fmt="cs8", taps=("lpf", 24000, 9600),
         calls=_calls(("xs", SEED + 30, 262144, "cf32"), ("xs", SEED + 31, 262144, "cs16"), ("xs", SEED + 32, 65538, "cf32"))),
    # ---- even tap count: the reversal quirk ---------------------------------------------------------
    dict(name="g12_even256", fs=10000000, D=100, fc=1234567, max_input=262144, fmt="cs16", taps=("hsinc", 256, 0.004),
         calls=_calls(("xs", SEED + 40, 131072, "cf32"), ("xs", SEED + 41, 131072, "cf32"))),
    # ---- config 5 (cf32-input extension): reference fed int16 q, HIP/oracle cf32 path fed q/32768 ---------------
    dict(name="g13_cf32_257", fs=10000000, D=100, fc=-2500000, max_input=524288, fmt="cf32", ref_fmt="cs16",
         taps=("hsinc", 257, 0.004),
         calls=_calls(("sinq", 0, 262144, "cf32"), ("xs", SEED + 50, 262144, "cf32"), ("xs", SEED + 51, 20002, "cf32"))),
    # ---- impulse responses: tap order, odd and even -------------------------------------------------------------
    dict(name="g14_impulse_odd", fs=48000, D=1, fc=0, max_input=64, fmt="cs16", taps=("list", [1, 2, 3, 4, 5, 6, 7], 8.0),
         calls=_calls(("impulse", 0, 32, "cf32"), ("impulse", 4, 32, "cs16"))),
    dict(name="g14_impulse_even", fs=48000, D=1, fc=0, max_input=64, fmt="cs16", taps=("list", [1, 2, 3, 4, 5, 6], 8.0),
         calls=_calls(("impulse", 0, 32, "cf32"), ("impulse", 4, 32, "cs16"))),
    dict(name="g14_impulse_shift", fs=48000, D=2, fc=6000, max_input=64, fmt="cs16", taps=("list", [1, 2, 3, 4, 5, 6, 7, 8], 16.0),
         calls=_calls(("impulse", 1, 32, "cf32"), ("impulse", 3, 32, "cf32"))),
    # ---- NCO drift: 100 default blocks on one filter, sample every 10th ----------------------------------------
    dict(name="g15_drift", fs=2016000, D=42, fc=-12000, max_input=262144, fmt="cu8", taps=("lpf", 24000, 9600),
         calls=[dict(gen="xs", a=SEED + 100 + k, n=262144, out="cf32", keep=(k % 10 == 0 or k == 99),
                     keep_n=(None if k == 99 else 64)) for k in range(100)]),
]

BY_NAME = {s["name"]: s for s in SCENARIOS}

# ---- "fast" fixtures: the reference's x86 Release-like build (AVX process_optimized_*, no phase renormalisation) on
# blocks 0-9 of three shapes (tests/golden/fast_<name>.npz; generator: make_golden.py make_fast)
FAST_SHAPES = ("g9_default", "g10_96k", "g11_t101")
FAST_BLOCKS, FAST_HEAD, FAST_TAIL = 10, 256, 64


def fast_block(sc, k):
    """Input block k of the fast fixtures: the scenario's first call shape with its own seed."""
    c0 = sc["calls"][0]
    return make_input(sc, dict(gen="xs", a=SEED + 7000 + 16 * FAST_SHAPES.index(sc["name"]) + k, n=c0["n"]))


# the long stream of the x86 build (tests/golden/x86_long_g9.npz): 400 blocks of the g9 shape cycling through the 16 fast-fixture
# blocks; sampled blocks keep their first X86_HEAD and last X86_TAIL outputs, the last block is kept whole
X86_LONG_BLOCKS, X86_LONG_SAMPLED, X86_HEAD, X86_TAIL = 400, (0, 9, 49, 99, 199, 299, 399), 128, 64


def make_taps(sc, lpf):
    """lpf(gain, fs, cutoff, tw) -> float32 taps (reference, oracle or HIP-side designer: they must agree)."""
    kind = sc["taps"][0]
    if kind == "lpf":
        return lpf(1.0, sc["fs"], sc["taps"][1], sc["taps"][2])
    if kind == "hsinc":
        return siggen.hamming_sinc(sc["taps"][1], sc["taps"][2])
    if kind == "list":
        return (np.asarray(sc["taps"][1], dtype=np.float32) / np.float32(sc["taps"][2])).astype(np.float32)
    raise ValueError(kind)


def make_input(sc, call):
    """Scalar-element array in the format the REFERENCE consumes for this scenario (sc['ref_fmt'] or sc['fmt'])."""
    fmt = sc.get("ref_fmt", sc["fmt"])
    g, a, n = call["gen"], call["a"], call["n"]
    if g == "ramp":
        return {"cu8": siggen.ramp_u8, "cs8": siggen.ramp_s8, "cs16": siggen.ramp_s16}[fmt](a, n)
    if g == "stair":
        assert fmt == "cu8"
        return siggen.staircase_u8(n)
    if g == "xs":
        return {"cu8": siggen.xs_u8, "cs8": siggen.xs_s8, "cs16": siggen.xs_s16}[fmt](a, n)
    if g == "sinq":
        assert fmt == "cs16"
        return np.round(siggen.sin_f32(a, n).astype(np.float64) * 32767.0).astype(np.int16)
    if g == "impulse":
        assert fmt == "cs16"
        x = np.zeros(n, dtype=np.int16)
        x[2 * a] = 16384
        x[2 * a + 1] = -8192
        return x
    raise ValueError(g)


def to_path_input(sc, x):
    """Convert reference-format input to what the oracle / HIP path consumes (differs only for the cf32 extension)."""
    if sc["fmt"] == "cf32" and sc.get("ref_fmt") == "cs16":
        return (x.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    return x
