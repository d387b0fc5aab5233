#!/usr/bin/env python3
"""timeline of one wide two-half mix launch from the dump written with XL_EXP_POLY_TRACE=<file> by a -DXL_TUNING build:
per wave [start, end, placement, first pass over]; 100 MHz clock"""
import sys
import numpy as np
h = np.fromfile(sys.argv[1], dtype=np.uint64)
nw = int(sys.argv[2])
w = h[4096:4096 + 4 * nw].reshape(nw, 4)
ok = w[:, 0] > 0
w = w[ok]
t0 = int(w[:, 0].min())
st = (w[:, 0].astype(np.int64) - t0) * 0.01
en = (w[:, 1].astype(np.int64) - t0) * 0.01
p1 = (w[:, 3].astype(np.int64) - t0) * 0.01
print(f"waves traced {len(w)}; launch span {en.max():.1f} us")
first = st < 2.0
print(f"first round: {first.sum()} waves start <2us; first pass over at med {np.median(p1[first]):.1f} (p10 {np.percentile(p1[first],10):.1f} p90 {np.percentile(p1[first],90):.1f}); end med {np.median(en[first]):.1f} p10 {np.percentile(en[first],10):.1f} p90 {np.percentile(en[first],90):.1f}")
sec = ~first
if sec.any():
    print(f"later rounds: {sec.sum()} waves; start med {np.median(st[sec]):.1f} (p10 {np.percentile(st[sec],10):.1f} p90 {np.percentile(st[sec],90):.1f}); first pass takes med {np.median((p1-st)[sec]):.1f}; whole wave med {np.median((en-st)[sec]):.1f}")
print(f"all: first pass takes med {np.median(p1-st):.1f}; remaining passes take med {np.median(en-p1):.1f} us")
sg = (w[:, 2].astype(np.int64) - t0) * 0.01
if (sg > 0).all() and (sg < en.max() + 1).all():  # (wide kernel, k-block-major form: slot 2 = A operands staged, slot 3 = products over)
    print(f"phases per wave (us): staging med {np.median(sg-st):.1f} p90 {np.percentile(sg-st,90):.1f} | products med {np.median(p1-sg):.1f} p90 {np.percentile(p1-sg,90):.1f} | stores med {np.median(en-p1):.1f} p90 {np.percentile(en-p1,90):.1f}")
    print(f"  first round only: staging {np.median((sg-st)[first]):.1f} products {np.median((p1-sg)[first]):.1f} stores {np.median((en-p1)[first]):.1f}")
    sys.exit(0)
hw = w[:, 2]
cu = [((int(v) >> 32) & 7, ((int(v) & 0xFFFFFFFF) >> 13) & 7, ((int(v) & 0xFFFFFFFF) >> 12) & 1, ((int(v) & 0xFFFFFFFF) >> 8) & 15) for v in hw]
from collections import Counter
cnt = Counter(cu)
print(f"CUs used {len(cnt)}; waves per CU min {min(cnt.values())} med {int(np.median(list(cnt.values())))} max {max(cnt.values())}")
# concurrency: how many waves are alive at a few instants
for t in (1, 5, 10, 15, 20, 25, 30, 35, 40, 45, 50, 55, 60):
    print(f"  t={t:2d} us: alive {int(((st <= t) & (en > t)).sum())}  in first pass {int(((st <= t) & (p1 > t)).sum())}")
