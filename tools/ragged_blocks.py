#!/usr/bin/env python3
"""tools/ragged_blocks.py -- GPU: what a call pattern with IRREGULAR block lengths costs (every wrong shape guess drops the
phase table tabulated ahead and redoes it with a stand-alone xl_nco_table_kernel launch, which claims whole SIMDs).  Native and
optimized mode, one block per call, constant 262144-byte blocks against lengths alternating 262144 / 262100 / 131072 bytes."""
import os

os.environ.setdefault("XL_TESTING", "1")  # (a tuning tool: the library honours XL_EXP_* only next to this)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import siggen  # noqa: E402
import sdr_server_amd as xl  # noqa: E402

FS, D, BLOCK = 2016000, 42, 262144
code, taps = xl.create_low_pass_filter(1.0, FS, 24000, 9600)
data = torch.from_numpy(siggen.xs_u8(99, BLOCK)).cuda()
print(f"{'mode':10s} {'clients':>7s} {'pattern':>10s} {'us/call':>9s}")
for mode in ("native", "optimized"):
    for n in (128, 1024, 4096):
        for pattern, lens in (("constant", [BLOCK]), ("ragged", [BLOCK, BLOCK - 44, BLOCK // 2])):
            eng = xl.BatchEngine(FS, "cu8", BLOCK)
            for c in range(n):
                eng.add_client(D, taps, -984000 + 1920 * (c % 1024) + 240 * (c // 1024))
            for k in range(6):
                eng.process_device_group(data.data_ptr(), lens[k % len(lens)], 1, mode, "engine")
            torch.cuda.synchronize()
            calls = 240
            t0 = time.perf_counter()
            for k in range(calls):
                eng.process_device_group(data.data_ptr(), lens[k % len(lens)], 1, mode, "engine")
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / calls
            # (normalise: a ragged cycle holds 2.5 blocks' worth of samples in 3 calls)
            per_block = dt * (3 / 2.5 if pattern == "ragged" else 1.0)
            print(f"{mode:10s} {n:7d} {pattern:>10s} {dt*1e6:9.2f}   (per 131072 samples: {per_block*1e6:.2f} us)", flush=True)
            eng.close()
