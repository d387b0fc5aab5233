#!/usr/bin/env python3
"""tools/inverse_midband.py -- GPU: the inverse launch's three kernels (option inverse_kernel = 3: LDS transform, 5: eight lanes per column,
6: the 32 x 4 cut) ALTERNATING IN ONE PROCESS on the launch sizes between the size rule's thresholds (xl_plan_rules.h: 2048 < tiles <=
8192 take the LDS transform) and one size either side -- box-to-box differences are larger than the differences between the kernels, so
only a same-process table says where the crossover lies.  Per (shape, kernel): the inverse launch's own duration (HIP events around it)
and the whole call's time per block, best and median of the rounds.  Prints a table (profiles/r06_inverse_midband.txt)."""
import argparse
import os

os.environ.setdefault("XL_TESTING", "1")  # (a tuning tool)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import siggen  # noqa: E402
import sdr_server_amd as xl  # noqa: E402

BLOCK = 262144
# (name, shape, clients, blocks per call)
SHAPES = [("1024 clients x 1 block", "server", 1024, 1), ("config 5, 1024 clients x 8 blocks", "config5", 1024, 8), ("4096 clients x 1 block", "server", 4096, 1),
          ("config 5, 2048 clients x 8 blocks", "config5", 2048, 8), ("1024 clients x 8 blocks", "server", 1024, 8), ("1536 clients x 8 blocks", "server", 1536, 8),
          ("2048 clients x 8 blocks", "server", 2048, 8)]


def run(shape, n, G, inv, data, blocks):
    if shape == "config5":
        fs, fmt, D, taps = 10000000, "cf32", 100, siggen.hamming_sinc(257, 0.004)
    else:
        fs, fmt, D = 2016000, "cu8", 42
        taps = xl.create_low_pass_filter(1.0, fs, 24000, 9600)[1]
    eng = xl.BatchEngine(fs, fmt, BLOCK, group_blocks=G)
    eng.set_option("inverse_kernel", inv)
    for c in range(n):
        eng.add_client(D, taps, (-4000000 + (8000000 // n) * c) if shape == "config5" else (-984000 + 1920 * (c % 1024) + 240 * (c // 1024)))
    calls = max(8, blocks // G)
    for _ in range(4):
        eng.process_device_group(data.data_ptr(), BLOCK, G, "optimized", "engine")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        eng.process_device_group(data.data_ptr(), BLOCK, G, "optimized", "engine")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (calls * G)
    eng.timing_stride(1)
    eng.timing(2)
    for _ in range(16):
        eng.process_device_group(data.data_ptr(), BLOCK, G, "optimized", "engine")
    torch.cuda.synchronize()
    n3, ms3 = eng.timing_polyphase()
    eng.timing(0)
    plan = eng.describe()
    eng.close()
    import re
    mm = re.search(r" V(\d+) M(\d+)", plan)
    V, M = int(mm.group(1)), int(mm.group(2))
    K = -(-BLOCK // 2 * G // D) + 1
    tiles = -(-K // V) * -(-n // 128) * 4
    return dt * 1e6, ms3[2] / max(n3, 1) * 1e3, tiles, M, plan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=320)
    ap.add_argument("--kernels", default="3,5,6")
    args = ap.parse_args()
    kernels = [int(k) for k in args.kernels.split(",")]
    d_u8 = torch.from_numpy(siggen.xs_u8(99, 8 * BLOCK)).cuda()
    d_f32 = torch.from_numpy((siggen.xs_s16(99, 8 * BLOCK).astype(np.float32) / np.float32(32768)).astype(np.float32)).cuda()
    print(f"# {xl.device_info()}; inverse_kernel 3 = LDS transform (xlp_inverse_kernel<128>), 5 = eight lanes per column (xl_inv8.hip), 6 = 32 x 4 cut (xl_inv32.hip)")
    print(f"# per cell: inverse launch us per CALL (HIP events) best / median of {args.rounds} alternating rounds | whole call us per BLOCK best")
    print(f"{'shape':36s} {'tiles':>6s} " + " ".join(f"{'inv=' + str(k):>28s}" for k in kernels) + "   fastest")
    for name, shape, n, G in SHAPES:
        res = {k: [] for k in kernels}
        tiles = M = 0
        for _ in range(args.rounds):
            for k in kernels:
                us_blk, inv_us, tiles, M, _ = run(shape, n, G, k, d_f32 if shape == "config5" else d_u8, args.blocks)
                res[k].append((inv_us, us_blk))
        if M != 128:
            print(f"{name:36s} (M = {M}: the LDS transform only)")
            continue
        cells, best = [], {}
        for k in kernels:
            inv = sorted(v[0] for v in res[k])
            blk = min(v[1] for v in res[k])
            best[k] = inv[0]
            cells.append(f"{inv[0]:7.1f} /{inv[len(inv) // 2]:7.1f} | {blk:6.2f}")
        fastest = min(best, key=best.get)
        second = sorted(best.values())[1]
        print(f"{name:36s} {tiles:6d} " + " ".join(f"{c:>28s}" for c in cells) + f"   {fastest} (by {100.0 * (second / best[fastest] - 1.0):.1f} %)", flush=True)


if __name__ == "__main__":
    main()
