#!/usr/bin/env python3
"""tools/measure_dropin.py -- latency of the drop-in process_* API (BASELINE config 2: 1 GPU, 1 client, latency-bound):
microseconds per 262144-byte cu8 block, 505 and 101 taps, both variants.  Torch-free on purpose: with torch's runtime
threads alive in the process the host-bound loop showed a 2x outlier (101 taps native) that a torch-free run never shows.
Prints one JSON object."""
import gc
import json
import os
import sys
import time

# The timed loops run with CPython's cyclic garbage collector off: a generation-2 collection inside the wrapper call is what
# rounds 2 and 3 first reported as "one call of 1-63 ms per few hundred" (tools/dropin_python_stall.py; the collector's
# triggers are allocation counts, so the stall landed in the same loop every run).  The C harness (tests/c/dropin_latency.c)
# measures the library without an interpreter around it.
gc.disable()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import siggen  # noqa: E402
import sdr_server_amd as xl  # noqa: E402

FS = 2016000
x = siggen.xs_u8(1, 262144)
res = {}
for rate, name in ((5, "505 taps"), (1, "101 taps")):
    taps = xl.create_low_pass_filter(1.0, FS, 24000, 48000 // rate)[1]
    for variant in ("native", "optimized"):
        f = xl.XlatingFilter(42, taps, -12000, FS, 262144)
        for _ in range(20):
            f.process(variant, "cu8", "cf32", x)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter()
            f.process(variant, "cu8", "cf32", x)
            ts.append(time.perf_counter() - t0)
        f.close()
        ts.sort()
        mean = sum(ts) / len(ts)
        # (p99 and max are reported next to the mean so that a stall would show)
        res[f"{name} {variant}"] = {"us_per_block": round(mean * 1e6, 1), "median_us": round(ts[len(ts) // 2] * 1e6, 1),
                                    "p99_us": round(ts[int(len(ts) * 0.99) - 1] * 1e6, 1), "max_us": round(ts[-1] * 1e6, 1),
                                    "Msps": round(131072 / mean / 1e6, 1)}
# the Q15 family (process_*_cu8_cs16; the server itself never calls it: dsp_worker.c:110-124 picks the cf32 family)
taps = xl.create_low_pass_filter(1.0, FS, 24000, 9600)[1]
f = xl.XlatingFilter(42, taps, -12000, FS, 262144)
for _ in range(10):
    f.process("native", "cu8", "cs16", x)
ts = []
for _ in range(100):
    t0 = time.perf_counter()
    f.process("native", "cu8", "cs16", x)
    ts.append(time.perf_counter() - t0)
f.close()
ts.sort()
res["505 taps cs16 (Q15) output"] = {"us_per_block": round(sum(ts) / len(ts) * 1e6, 1), "median_us": round(ts[len(ts) // 2] * 1e6, 1),
                                      "Msps": round(131072 / (sum(ts) / len(ts)) / 1e6, 1)}
res["note"] = "timed through the ctypes wrapper with the garbage collector disabled; from C: config1_dropin_latency_from_C_10000_calls"
print(json.dumps(res))
