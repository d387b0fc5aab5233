#!/usr/bin/env python3
"""tools/group_sweep.py -- GPU: time per BLOCK vs blocks per call (G), client count, variant, transform length.
Run on the GPU box:  python tools/group_sweep.py [--clients 128,1024] [--groups 1,2,4,8] [--modes optimized,native] [--m 0,128,256]"""
import argparse
import os

os.environ.setdefault("XL_TESTING", "1")  # (a tuning tool: the library honours XL_EXP_* only next to this)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clients", default="128,256,1024,4096")
    ap.add_argument("--groups", default="1,2,4,8")
    ap.add_argument("--modes", default="optimized")
    ap.add_argument("--m", default="0")
    ap.add_argument("--rate", type=int, default=5)
    ap.add_argument("--decimations", default="42", help="client decimations to time (input rate = 48 kHz x decimation), e.g. 42,50,64")
    ap.add_argument("--blocks", type=int, default=320, help="blocks in the timed region")
    ap.add_argument("--poly3", action="store_true", help="also time the three polyphase launches separately")
    ap.add_argument("--slices", default="", help="forward | inverse boundary of the in-launch NCO role, in 1/65536 of a call")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (repeatable), e.g. mix_kernel=2")
    ap.add_argument("--shape", default="server", choices=["server", "config5"], help="config5: BASELINE configs[4] -- cf32 input at 10 Msps, D = 100, 257 explicit taps")
    ap.add_argument("--engine-stream", type=int, default=1, help="1: XL_STREAM_ENGINE (the engine's own, CU-masked compute stream); 0: torch's stream")
    args = ap.parse_args()
    gmax = max(int(g) for g in args.groups.split(","))
    if args.shape == "config5":
        data = torch.from_numpy((siggen.xs_s16(99, gmax * BLOCK).astype(np.float32) / np.float32(32768)).astype(np.float32)).cuda()
    else:
        data = torch.from_numpy(siggen.xs_u8(99, gmax * BLOCK)).cuda()
    st = torch.cuda.current_stream()
    sarg = "engine" if args.engine_stream else st.cuda_stream
    print(f"{'mode':10s} {'M':>4s} {'clients':>7s} {'G':>2s} {'us/block':>9s} {'kern us/blk':>11s} {'Msps':>10s}   plan / launches us per block")
    for dec in ([100] if args.shape == "config5" else [int(v) for v in args.decimations.split(",")]):
        sweep(args, dec, data, sarg)


def sweep(args, D, data, sarg):
    FS = 48000 * D
    code, taps = xl.create_low_pass_filter(1.0, FS, 24000, 48000 // args.rate)
    fmt = "cu8"
    if args.shape == "config5":
        FS, fmt, taps = 10000000, "cf32", siggen.hamming_sinc(257, 0.004)
    for mode in args.modes.split(","):
        for m in [int(v) for v in args.m.split(",")]:
            for n in [int(c) for c in args.clients.split(",")]:
                for G in [int(g) for g in args.groups.split(",")]:
                    # (tuning knobs are no engine options: XL_EXP_* environment variables read when the engine is created)
                    tuning = {"mix_passes_per_workgroup": "XL_EXP_MIX_PP", "polyphase_min_clients": "XL_EXP_POLY_MIN", "tile_height": "XL_EXP_H",
                              "riders": "XL_EXP_RIDERS", "riders_min_workgroups": "XL_EXP_RIDERS_MIN", "nco_calls_per_launch": "XL_EXP_CHAIN_CALLS",
                              "nco_slice": "XL_EXP_NCO_SLICE"}
                    opts = [kv.split("=") for kv in args.opt]
                    if args.slices:
                        opts.append(("nco_slice", args.slices))
                    for name, val in opts:
                        if name in tuning:
                            os.environ[tuning[name]] = val
                    eng = xl.BatchEngine(FS, fmt, BLOCK, group_blocks=G)
                    if m:
                        eng.set_option("polyphase_m", m)
                    for name, val in opts:
                        if name not in tuning:
                            eng.set_option(name, int(val))
                    for c in range(n):
                        if args.shape == "config5":
                            eng.add_client(D, taps, -4000000 + (8000000 // n) * c)
                        else:
                            eng.add_client(D, taps, -984000 + 1920 * (c % 1024) + 240 * (c // 1024))
                    calls = max(4, args.blocks // G)
                    for k in range(4):
                        eng.process_device_group(data.data_ptr(), BLOCK, G, mode, sarg)
                    torch.cuda.synchronize()
                    eng.timing_stride(2)
                    eng.timing(1)
                    t0 = time.perf_counter()
                    for k in range(calls):
                        eng.process_device_group(data.data_ptr(), BLOCK, G, mode, sarg)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / (calls * G)
                    nt, fir, nco = eng.timing_read()
                    eng.timing(0)
                    extra = ""
                    if args.poly3 and "polyphase: none" not in eng.describe() and mode == "optimized":
                        eng.timing_stride(1)
                        eng.timing(2)
                        for k in range(8):
                            eng.process_device_group(data.data_ptr(), BLOCK, G, mode, sarg)
                        torch.cuda.synchronize()
                        n3, ms3 = eng.timing_polyphase()
                        eng.timing(0)
                        if n3:
                            extra = "fwd %.1f mix %.1f inv %.1f" % tuple(v / n3 / G * 1e3 for v in ms3)
                    plan = eng.describe()
                    if os.environ.get("XL_EXP_CHAIN_STATS"):
                        import ctypes as C
                        nwg = (n + 63) // 64
                        buf = (C.c_ulonglong * (4 * nwg))()
                        xl.lib().xlating_batch_debug_chain_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
                        if xl.lib().xlating_batch_debug_chain_stats(eng.h, buf, nwg) == 0:
                            cyc = [buf[4 * i] for i in range(nwg)]; tick = [buf[4 * i + 1] for i in range(nwg)]; ent = buf[2]
                            st = [buf[4 * i + 3] for i in range(nwg)]
                            ent = max(ent, 1)
                            big = (C.c_ulonglong * (4 * 4096))()
                            if os.environ.get("XL_EXP_CHAIN_TIMELINE") and xl.lib().xlating_batch_debug_chain_stats(eng.h, big, 4096) == 0:
                                tl = [big[8192 + i] for i in range(8)]
                                extra += "  launch timeline wg0 (us from entry): " + " ".join("%.1f" % (v / 100.0) for v in tl[1:])
                            extra += "  chain: %.1f cycles/step, %.2f ns/step, clock %.2f GHz, start spread %.1f us" % (
                                sum(cyc) / nwg / (ent * 16), sum(tick) / nwg * 10.0 / (ent * 16), sum(cyc) / max(sum(tick), 1) / 10.0, (max(st) - min(st)) / 100.0)
                    eng.close()
                    kern = fir / max(nt, 1) / G * 1e3
                    print(f"{mode:10s} {m:4d} {n:7d} {G:2d} {dt*1e6:9.2f} {kern:11.2f} {n*131072/dt/1e6:10.0f}   {plan.split('|')[2].strip()[:60]} {extra}", flush=True)


if __name__ == "__main__":
    main()
