#!/usr/bin/env python3
"""tools/dropin_python_stall.py -- GPU: where the "1-35 ms call once per thousand" of round 2's drop-in numbers came from.
The same process_optimized_cu8_cf32 loop measured three ways: the ctypes call alone (no numpy copy), the wrapper's
process() with the garbage collector enabled, and with it disabled.  The C harness (tests/c/dropin_latency.c) shows no
call over 0.2 ms in 10 000; whatever shows up here beyond that is the Python host, not the library."""
import ctypes as C
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import siggen  # noqa: E402
import sdr_server_amd as xl  # noqa: E402

FS, N = 2016000, 10000
x = siggen.xs_u8(1, 262144)
taps = xl.create_low_pass_filter(1.0, FS, 24000, 9600)[1]
res = {}


def stats(ts):
    s = sorted(ts)
    return {"median_us": round(s[len(s) // 2] * 1e6, 1), "p999_us": round(s[int(len(s) * 0.999)] * 1e6, 1), "max_us": round(s[-1] * 1e6, 1),
            "calls_over_1ms": sum(1 for t in s if t > 1e-3)}


for name in ("wrapper, gc enabled", "wrapper, gc disabled", "bare ctypes call, gc enabled"):
    f = xl.XlatingFilter(42, taps, -12000, FS, 262144)
    for _ in range(50):
        f.process("optimized", "cu8", "cf32", x)
    if "disabled" in name:
        gc.collect()
        gc.disable()
    ts = []
    if name.startswith("bare"):
        fn = xl.lib().process_optimized_cu8_cf32
        p, n = C.POINTER(C.c_float)(), C.c_size_t(0)
        xp = x.ctypes.data_as(C.POINTER(C.c_uint8))
        for _ in range(N):
            t0 = time.perf_counter()
            fn(xp, x.size, C.byref(p), C.byref(n), f.h)
            ts.append(time.perf_counter() - t0)
    else:
        for _ in range(N):
            t0 = time.perf_counter()
            f.process("optimized", "cu8", "cf32", x)
            ts.append(time.perf_counter() - t0)
    gc.enable()
    f.close()
    res[name] = stats(ts)
print(json.dumps(res))
