#!/usr/bin/env python3
"""Generate the committed golden fixtures for the xlating hot path.

Run in the BUILD container only (needs /root/reference and oracle/_ref/libref_canon.so, which
oracle/Makefile compiles from the unmodified reference sources):

    python tests/golden/make_golden.py

Outputs (data only -- numbers, never reference source text):
  tests/golden/ref_test_vectors.json   the expected-value arrays held by the reference's own tests
                                       (test/test_xlating.c, test/test_tcp_server.c, test/test_lpf.c),
                                       parsed out of the C initialisers as numbers.
  tests/golden/live_<name>.npz         outputs of the unmodified reference (canonical flags) for the
                                       scenarios in scenarios.py.
  tests/golden/fast_<name>.npz         outputs of the unmodified reference's x86 Release-like build (oracle/_ref/
                                       libref_fast.so: -O3 -ffast-math -mavx2 -mfma, i.e. the hand-written AVX
                                       process_optimized_* of xlating.c:271-348, which does NOT renormalise the phase,
                                       :338-339) for blocks 0-9 of the g9 / g10 / g11 shapes: what an x86 server's
                                       "optimized" setting computes.  Head and tail of every block + block 9 in full.
  tests/golden/avx_g11_t101.npz        blocks 0-9 of the g11 shape from the reference built with -mavx but WITHOUT FMA (libref_avx.so).
  tests/golden/x86_long_g9.npz         the same AVX build over 400 blocks of the g9 shape, sampled (pins the no-renormalisation mode).
  tests/golden/fast_divergence.json    how far the reference's own two builds drift apart over a long stream
                                       (canonical native vs AVX optimized; max |d| / max |y| per block).
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import scenarios  # noqa: E402
from pyoracle import RefLib, build  # noqa: E402

REF = "/root/reference"


def parse_arrays(path):
    """-> {function_name: {array_name: [numbers]}} for every `const float|int16_t name[] = {...};`"""
    src = open(path).read()
    out = {}
    fn_pat = re.compile(r"^void\s+(\w+)\s*\(\s*\)\s*\{", re.M)
    fns = [(m.start(), m.group(1)) for m in fn_pat.finditer(src)]
    arr_pat = re.compile(r"const\s+(float|int16_t)\s+(\w+)\[\]\s*=\s*\{([^}]*)\}", re.S)
    for m in arr_pat.finditer(src):
        owner = None
        for pos, name in fns:
            if pos < m.start():
                owner = name
        vals = [v.strip() for v in m.group(3).replace("\n", " ").split(",") if v.strip()]
        if m.group(1) == "float":
            nums = [float(v.rstrip("fF")) for v in vals]
        else:
            nums = [int(v) for v in vals]
        out.setdefault(owner, {})[m.group(2)] = nums
    return out


def main():
    build(ref=True)
    vec = {
        "_comment": "expected-value arrays of the reference's own tests for the xlating/lpf path (numbers only)",
        "test_xlating.c": parse_arrays(f"{REF}/test/test_xlating.c"),
        "test_tcp_server.c": {k: v for k, v in parse_arrays(f"{REF}/test/test_tcp_server.c").items()
                              if k in ("test_rtlsdr", "test_airspy", "test_hackrf")},
        "test_lpf.c": parse_arrays(f"{REF}/test/test_lpf.c"),
    }
    with open(os.path.join(HERE, "ref_test_vectors.json"), "w") as f:
        json.dump(vec, f, indent=0, separators=(",", ":"))
    total = 0
    for sc in scenarios.SCENARIOS:
        taps = scenarios.make_taps(sc, lpf=lambda *a: RefLib.lpf(*a)[1])
        ref = RefLib(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"])
        arrays = {"taps": taps}
        for ci, call in enumerate(sc["calls"]):
            x = scenarios.make_input(sc, call)
            fmt = sc["ref_fmt"] if "ref_fmt" in sc else sc["fmt"]
            y = ref.process(fmt, x, call["out"])
            if call.get("keep", True):
                keep = call.get("keep_n")
                arrays[f"y{ci}"] = y if keep is None else y[:keep]
            arrays[f"n{ci}"] = np.int64(len(y))
        ref.close()
        p = os.path.join(HERE, f"live_{sc['name']}.npz")
        np.savez_compressed(p, **arrays)
        sz = os.path.getsize(p)
        total += sz
        print(f"{sc['name']:28s} T={taps.size:5d} D={sc['D']:4d} calls={len(sc['calls']):4d} -> {sz/1024:.1f} KiB")
    print(f"total {total/1024:.1f} KiB")
    make_fast()


FAST_SHAPES, FAST_BLOCKS, FAST_HEAD, FAST_TAIL = scenarios.FAST_SHAPES, scenarios.FAST_BLOCKS, scenarios.FAST_HEAD, scenarios.FAST_TAIL
fast_block = scenarios.fast_block


def make_fast():
    if not RefLib.available("fast") or " avx2 " not in (" " + open("/proc/cpuinfo").read().replace("\n", " ") + " "):
        print("fast fixtures skipped: libref_fast.so / AVX2 not available here")
        return
    assert RefLib.simd_status("fast") == "AVX", RefLib.simd_status("fast")
    for name in FAST_SHAPES:
        sc = scenarios.BY_NAME[name]
        taps = scenarios.make_taps(sc, lpf=lambda *a: RefLib.lpf(*a)[1])
        fast = RefLib(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"], flavour="fast", variant="optimized")
        arrays = {"taps": taps}
        for k in range(FAST_BLOCKS):
            y = fast.process(sc["fmt"], fast_block(sc, k), "cf32")
            arrays[f"n{k}"] = np.int64(len(y))
            if k == FAST_BLOCKS - 1:
                arrays[f"y{k}"] = y
            else:
                arrays[f"head{k}"] = y[:FAST_HEAD]
                arrays[f"tail{k}"] = y[-FAST_TAIL:]
        fast.close()
        p = os.path.join(HERE, f"fast_{name}.npz")
        np.savez_compressed(p, **arrays)
        print(f"fast_{name:24s} T={taps.size:5d} -> {os.path.getsize(p)/1024:.1f} KiB")
    # how far the reference's own builds drift apart (g9 shape): canonical native renormalises every call, AVX never does
    sc = scenarios.BY_NAME["g9_default"]
    taps = scenarios.make_taps(sc, lpf=lambda *a: RefLib.lpf(*a)[1])
    canon = RefLib(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"], flavour="canon", variant="native")
    fast = RefLib(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"], flavour="fast", variant="optimized")
    div = {}
    for k in range(400):
        x = fast_block(sc, k % 16)
        a, b = canon.process("cu8", x), fast.process("cu8", x)
        if k in (0, 1, 4, 9, 19, 49, 99, 199, 399):
            div[str(k)] = float(np.abs(a.astype(np.complex128) - b).max() / np.abs(a).max())
    canon.close()
    fast.close()
    json.dump({"_comment": "max|canonical native - AVX optimized| / max|y| per block of the g9 shape (505 taps, D=42, 262144-byte cu8 "
                           "blocks); both are the UNMODIFIED reference, oracle/_ref/libref_canon.so vs libref_fast.so",
               "block": div}, open(os.path.join(HERE, "fast_divergence.json"), "w"), indent=1)
    print("reference canon-vs-fast divergence:", div)
    make_x86_long()
    make_avx_g11()


def make_avx_g11():
    """tests/golden/avx_g11_t101.npz: blocks 0-9 of the g11 shape from the reference built WITHOUT FMA (oracle/_ref/libref_avx.so:
    the reference's own Release flags -O3 -ffast-math, + -mavx) -- the shape on which the FMA build's contracted phase step
    (fast_g11_t101.npz) and the plain one part ways within a few blocks.  Pins XL_MODE_OPTIMIZED_X86 (plain step)."""
    if not RefLib.available("avx"):
        return
    sc = scenarios.BY_NAME["g11_t101"]
    taps = scenarios.make_taps(sc, lpf=lambda *a: RefLib.lpf(*a)[1])
    avx = RefLib(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"], flavour="avx", variant="optimized")
    assert RefLib.simd_status("avx") == "AVX"
    arrays = {"taps": taps}
    for k in range(FAST_BLOCKS):
        y = avx.process(sc["fmt"], fast_block(sc, k), "cf32")
        arrays[f"n{k}"] = np.int64(len(y))
        if k == FAST_BLOCKS - 1:
            arrays[f"y{k}"] = y
        else:
            arrays[f"head{k}"] = y[:FAST_HEAD]
            arrays[f"tail{k}"] = y[-FAST_TAIL:]
    avx.close()
    p = os.path.join(HERE, "avx_g11_t101.npz")
    np.savez_compressed(p, **arrays)
    print(f"avx_g11_t101 -> {os.path.getsize(p)/1024:.1f} KiB")


def make_x86_long():
    """tests/golden/x86_long_g9.npz: the reference's AVX process_optimized_cu8_cf32 (oracle/_ref/libref_fast.so) over a LONG
    stream -- 400 server-default blocks, where its never-renormalised phasor has drifted by 1.5e-3 in amplitude -- sampled at
    a few blocks.  Pins XL_MODE_OPTIMIZED_X86 / xlating_set_optimized_x86 outright (<= 1e-5, no scale factor)."""
    sc = scenarios.BY_NAME["g9_default"]
    taps = scenarios.make_taps(sc, lpf=lambda *a: RefLib.lpf(*a)[1])
    fast = RefLib(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"], flavour="fast", variant="optimized")
    arrays = {"taps": taps}
    for k in range(scenarios.X86_LONG_BLOCKS):
        y = fast.process(sc["fmt"], fast_block(sc, k % 16), "cf32")
        if k in scenarios.X86_LONG_SAMPLED:
            arrays[f"n{k}"] = np.int64(len(y))
            if k == scenarios.X86_LONG_BLOCKS - 1:
                arrays[f"y{k}"] = y
            else:
                arrays[f"head{k}"] = y[:scenarios.X86_HEAD]
                arrays[f"tail{k}"] = y[-scenarios.X86_TAIL:]
    fast.close()
    p = os.path.join(HERE, "x86_long_g9.npz")
    np.savez_compressed(p, **arrays)
    print(f"x86_long_g9 -> {os.path.getsize(p)/1024:.1f} KiB")


if __name__ == "__main__":
    main()
