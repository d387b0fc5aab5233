#!/usr/bin/env python3
"""tools/poly_check.py -- GPU: polyphase path vs direct path on the same stream (accuracy + time)."""
import os, sys, time
os.environ.setdefault("XL_TESTING", "1")  # (a tuning tool: the library honours XL_EXP_* only next to this)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, siggen, sdr_server_amd as xl

def run(n, poly, rate=5, steps=100, mode="optimized"):
    os.environ["XL_EXP_POLY"] = poly
    code, taps = xl.create_low_pass_filter(1.0, bench.FS, bench.RATE // 2, bench.RATE // rate)
    eng = xl.BatchEngine(bench.FS, "cu8", bench.BLOCK_BYTES)
    for c in range(n):
        eng.add_client(bench.D, taps, bench.client_center_freq(c))
    desc = eng.describe()
    blocks = [torch.from_numpy(b).cuda() for b in [siggen.xs_u8(123 + k, bench.BLOCK_BYTES) for k in range(4)]]
    st = torch.cuda.current_stream()
    outs = []
    for k in range(6):
        eng.process_device(blocks[k % 4].data_ptr(), bench.BLOCK_BYTES, mode, st.cuda_stream)
        if k >= 4:
            eng.fetch(); outs.append(np.concatenate([eng.output(c) for c in range(0, n, max(1, n // 32))]))
    torch.cuda.synchronize(); eng.timing(True)
    t0 = time.perf_counter()
    for k in range(steps):
        eng.process_device(blocks[k % 4].data_ptr(), bench.BLOCK_BYTES, mode, st.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    nt, fir, nco = eng.timing_read()
    eng.close()
    return desc, dt, fir / max(nt, 1), outs

rates = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "5,1").split(",")]
for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024").split(",")]:
    for rate in rates:
        d0, t0, f0, o0 = run(n, "0", rate)
        d1, t1, f1, o1 = run(n, "1", rate)
        err = max(float(np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max() / np.abs(a).max()) for a, b in zip(o0, o1))
        print(f"clients {n} rate {rate} ({d1.split(' T')[1].split(' ')[0] if ' T' in d1 else '?'} taps): direct {t0*1e3:.4f} ms (kernels {f0:.4f})  polyphase {t1*1e3:.4f} ms (kernels {f1:.4f})  "
              f"x{t0/t1:.2f}  max rel diff {err:.2e}  Msps {n*bench.S/t1/1e6:.0f}")
        print("   ", d1)
