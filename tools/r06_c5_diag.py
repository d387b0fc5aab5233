#!/usr/bin/env python3
"""diagnostic: config 5 block 0 (tone input) on POISONED device memory (NaN-filled, freed before the engine allocates): which outputs depend
on memory the engine never wrote"""
import os, sys
os.environ.setdefault("XL_TESTING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import siggen, sdr_server_amd as xl
from pyoracle import population
taps = siggen.hamming_sinc(257, 0.004); nsamp = 131072
def poison():
    bufs = [torch.full((256 * 1024 * 1024,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(24)]  # 24 GB of NaN
    torch.cuda.synchronize(); del bufs; torch.cuda.empty_cache()
for n in (256,):
    fcs = [-4900000 + (9800000 // n) * c for c in range(n)]
    x = np.concatenate([siggen.sin_f32(0, 2 * nsamp), (siggen.xs_s16(91, 2 * nsamp).astype(np.float32) / np.float32(32768))]).astype(np.float32)
    want0 = population(100, taps, fcs, 10000000, 2 * nsamp, "cf32", x[:2 * nsamp], 1)
    for mix in (0,):
        poison()
        eng = xl.BatchEngine(10000000, "cf32", 2 * nsamp)
        if mix: eng.set_option("mix_kernel", mix)
        ids = [eng.add_client(100, taps, fc) for fc in fcs]
        eng.process_host(x[:2 * nsamp], "optimized"); eng.fetch()
        plan = eng.describe(); V = int(plan.split(" V")[1].split()[0])
        bad = {}
        for c in range(n):
            got = eng.output(ids[c]); w = want0[c]
            d = np.abs(got.astype(np.complex128) - w)
            d = np.where(np.isfinite(d), d, np.inf)
            for s in range(-(-len(d) // V)):
                if d[s * V:(s + 1) * V].max() > 1e-5 * np.abs(w).max():
                    bad.setdefault(s, []).append(c)
        print(f"n={n} mix={mix} {plan.split('|')[2].strip()[:70]}: failing (segment: clients) " + "; ".join(f"{s}: {len(v)} [{' '.join(map(str, v[:40]))}]" for s, v in sorted(bad.items())))
        for s_, v in sorted(bad.items())[:1]:
            c = v[0]
            got = eng.output(ids[c]); w = want0[c]
            seg = slice(s_ * V, (s_ + 1) * V)
            d = got[seg].astype(np.complex128) - w[seg]
            print(f"   client {c} segment {s_}: nan count {int(np.isnan(got[seg]).sum())}; max|d| {np.nanmax(np.abs(d)):.3g} vs max|y| seg {np.abs(w[seg]).max():.3g}; spectrum of d (|DFT| top bins): " +
                  " ".join(f"{k}:{a:.2g}" for k, a in sorted(enumerate(np.abs(np.fft.fft(np.nan_to_num(d), 256 if V > 128 else 128)) / V), key=lambda t: -t[1])[:6]))
        eng.close()
