#!/usr/bin/env python3
"""tools/sweep.py -- GPU tuning sweep: FIR/NCO kernel time vs client count, tap count and variant.
Run on the GPU box:  python tools/sweep.py [--clients 256,512,...] [--rates 5,1] [--modes optimized,native]"""
import argparse
import os

os.environ.setdefault("XL_TESTING", "1")  # (a tuning tool: the library honours XL_EXP_* only next to this)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
import siggen  # noqa: E402
import sdr_server_amd as xl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clients", default="64,256,512,768,960,1024,1536,2048,4096")
    ap.add_argument("--rates", default="5,1")
    ap.add_argument("--modes", default="optimized")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--depths", default="1")
    args = ap.parse_args()
    blocks = [torch.from_numpy(b).cuda() for b in [siggen.xs_u8(123 + k, bench.BLOCK_BYTES) for k in range(4)]]
    stream = torch.cuda.current_stream()
    depths = [int(d) for d in args.depths.split(",")]
    print(f"{'mode':10s} {'dp':>2s} {'taps':>5s} {'clients':>7s} {'step_ms':>9s} {'fir_ms':>9s} {'nco_ms':>9s} {'TFLOP/s':>8s} {'algGB/s':>8s} {'Msps':>10s}")
    for mode in args.modes.split(","):
        for rate in [int(r) for r in args.rates.split(",")]:
            code, taps = xl.create_low_pass_filter(1.0, bench.FS, bench.RATE // 2, bench.RATE // rate)
            for n, depth in [(int(c), d) for c in args.clients.split(",") for d in depths]:
                eng = xl.BatchEngine(bench.FS, "cu8", bench.BLOCK_BYTES)
                for c in range(n):
                    eng.add_client(bench.D, taps, bench.client_center_freq(c))
                for k in range(5):
                    eng.process_device(blocks[k % 4].data_ptr(), bench.BLOCK_BYTES, mode, stream.cuda_stream)
                torch.cuda.synchronize()
                eng.timing(True)
                t0 = time.perf_counter()
                for k in range(args.steps):
                    eng.process_device(blocks[k % 4].data_ptr(), bench.BLOCK_BYTES, mode, stream.cuda_stream)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / args.steps
                nt, fir, nco = eng.timing_read()
                eng.close()
                fir = fir / nt if nt else dt * 1e3
                nco /= max(nt, 1)
                units = n * bench.S
                tf = units * bench.flops_per_unit(taps.size, bench.D) / (fir * 1e-3) / 1e12
                gb = units * bench.algorithmic_bytes_per_unit(bench.D) / (fir * 1e-3) / 1e9
                print(f"{mode:10s} {depth:2d} {taps.size:5d} {n:7d} {dt*1e3:9.4f} {fir:9.4f} {nco:9.4f} {tf:8.2f} {gb:8.1f} {units/dt/1e6:10.0f}", flush=True)


if __name__ == "__main__":
    main()
