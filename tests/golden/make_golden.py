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


if __name__ == "__main__":
    main()
