#!/usr/bin/env python3
"""tools/trace_analyze.py <trace.bin> -- timeline of one FIR launch from per-wave wall_clock64 stamps (100 MHz)."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 4).astype(np.int64)  # [block][wave][stamp]
blocks = np.arange(t.shape[0])
valid = t[:, :, 3] > 0
v = t[valid]
xcd = np.repeat(blocks[:, None] % 8, 4, axis=1)[valid]
t0 = v[:, 0].min()
us = (v - t0) / 100.0
print(f"waves traced: {len(v)}   launch span: {us[:,3].max():.1f} us")
for name, col in (("entry", 0), ("staged", 1), ("filtered", 2), ("stored", 3)):
    c = us[:, col]
    print(f"  {name:9s} min {c.min():7.1f}  p10 {np.percentile(c,10):7.1f}  med {np.median(c):7.1f}  p90 {np.percentile(c,90):7.1f}  max {c.max():7.1f}")
d = np.diff(us, axis=1)
for name, col in (("staging", 0), ("fir loop", 1), ("epilogue", 2)):
    c = d[:, col]
    print(f"  dur {name:9s} min {c.min():7.1f}  med {np.median(c):7.1f}  p90 {np.percentile(c,90):7.1f}  max {c.max():7.1f}")
# concurrency over time
edges = np.linspace(0, us[:, 3].max(), 21)
print("  resident waves over time (per 5% of the span):")
print("   ", " ".join(f"{int(((us[:,0] <= e) & (us[:,3] > e)).sum()):5d}" for e in edges[:-1]))
late = us[:, 0] > np.percentile(us[:, 0], 50) + 5
print(f"  waves starting >5us after the median start: {late.sum()}  (their start med {np.median(us[late,0]) if late.any() else 0:.1f} us)")
for k in range(8):
    m = xcd == k
    if m.any():
        print(f"  xcd {k}: waves {m.sum():5d}  first start {us[m,0].min():6.1f}  last end {us[m,3].max():6.1f}")
# ---- which workgroups stage slowly / finish last?
bidx = np.repeat(blocks[:, None], 4, axis=1)[valid]
widx = np.repeat(np.arange(4)[None, :], t.shape[0], axis=0)[valid]
stg = d[:, 0]
order = np.argsort(-stg)[:24]
print("  slowest staging (block, wave, xcd, staging us, loop us, end us):")
print("   ", " ".join(f"({bidx[i]},{widx[i]},{bidx[i]%8},{stg[i]:.0f},{d[i,1]:.0f},{us[i,3]:.0f})" for i in order))
slow = stg > 3 * np.median(stg)
print(f"  slow-staging waves: {slow.sum()}  distinct blocks: {len(set(bidx[slow]))}  block idx range: {bidx[slow].min() if slow.any() else 0}..{bidx[slow].max() if slow.any() else 0}")
h, e = np.histogram(bidx[slow], bins=16, range=(0, t.shape[0]))
print("  slow-staging waves by block-index sixteenth:", h.tolist())
last = np.argsort(-us[:, 3])[:16]
print("  last finishing (block, wave, start, staged, end):", " ".join(f"({bidx[i]},{widx[i]},{us[i,0]:.0f},{us[i,1]:.0f},{us[i,3]:.0f})" for i in last))
