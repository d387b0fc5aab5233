#!/usr/bin/env python3
"""tools/trace_analyze.py <trace.bin> -- timeline of one FIR launch from per-wave stamps written by xl_fir_kernel when
XL_EXP_TRACE is set: 6 uint64 per wave = wall_clock64 (100 MHz) at entry / staged / filtered / stored, HW_ID, XCC_ID.
NCO-role waves have stamps 1 and 2 equal to 0."""
import sys
import numpy as np

t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 6).astype(np.int64)  # [block][wave][word]
blk = np.repeat(np.arange(t.shape[0])[:, None], 4, axis=1)
wav = np.repeat(np.arange(4)[None, :], t.shape[0], axis=0)
live = t[:, :, 3] > 0
nco = live & (t[:, :, 1] == 0)
fir = live & ~nco
t0 = t[live][:, 0].min()


def hw(x):  # HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13(+)
    return (x >> 4) & 3, (x >> 8) & 15, (x >> 12) & 1, (x >> 13) & 7


v = t[fir]
us = (v[:, :4] - t0) / 100.0
print(f"FIR waves: {len(v)}   NCO-role waves: {nco.sum()}   launch span: {us[:,3].max():.1f} us")
for name, col in (("entry", 0), ("staged", 1), ("filtered", 2), ("stored", 3)):
    c = us[:, col]
    print(f"  {name:9s} min {c.min():7.1f}  p10 {np.percentile(c,10):7.1f}  med {np.median(c):7.1f}  p90 {np.percentile(c,90):7.1f}  max {c.max():7.1f}")
d = np.diff(us, axis=1)
for name, col in (("staging", 0), ("fir loop", 1), ("epilogue", 2)):
    c = d[:, col]
    print(f"  dur {name:9s} min {c.min():7.1f}  med {np.median(c):7.1f}  p90 {np.percentile(c,90):7.1f}  max {c.max():7.1f}")
edges = np.linspace(0, us[:, 3].max(), 21)
print("  resident FIR waves over time (per 5% of the span):")
print("   ", " ".join(f"{int(((us[:,0] <= e) & (us[:,3] > e)).sum()):5d}" for e in edges[:-1]))
# ---- placement: key = (xcc, se, sh, cu, simd)
simd, cu, sh, se = hw(v[:, 4])
xcc = v[:, 5] & 15
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
skey = key * 4 + simd
if nco.any():
    n = t[nco]
    nus = (n[:, [0, 3]] - t0) / 100.0
    print(f"  NCO role: start med {np.median(nus[:,0]):.1f}  end med {np.median(nus[:,1]):.1f}  max {nus[:,1].max():.1f} us")
    ns, ncu, nsh, nse = hw(n[:, 4])
    nkey = (((n[:, 5] & 15) * 8 + nse) * 2 + nsh) * 16 + ncu
    nskey = set((nkey * 4 + ns).tolist())
    on = np.array([k in nskey for k in skey])
    oncu = np.isin(key, nkey)
    for lab, m in (("same SIMD as an NCO wave", on), ("same CU, other SIMD", oncu & ~on), ("CU without NCO wave", ~oncu)):
        if m.any():
            print(f"  FIR waves on {lab:26s}: {m.sum():5d}  loop med {np.median(d[m,1]):6.1f}  p90 {np.percentile(d[m,1],90):6.1f}  end med {np.median(us[m,3]):6.1f}  end max {us[m,3].max():6.1f}")
cus, cnt = np.unique(key, return_counts=True)
print(f"  CUs used: {len(cus)}  FIR waves per CU: min {cnt.min()} med {int(np.median(cnt))} max {cnt.max()}   "
      f"hist {dict(zip(*np.unique(cnt, return_counts=True)))}")
sk, sc = np.unique(skey, return_counts=True)
print(f"  FIR waves per SIMD: hist {dict(zip(*np.unique(sc, return_counts=True)))}")
endcu = {k: us[key == k, 3].max() for k in cus}
byload = {}
for k, c in zip(cus, cnt):
    byload.setdefault(int(c), []).append(endcu[k])
print("  last end per CU by FIR waves on it:", {c: f"med {np.median(x):.0f} max {max(x):.0f} (n={len(x)})" for c, x in sorted(byload.items())})
last = np.argsort(-us[:, 3])[:12]
bl, wv = blk[fir], wav[fir]
print("  last finishing (block, wave, xcc, cu, simd, staged, end):", " ".join(
    f"({bl[i]},{wv[i]},{xcc[i]},{se[i]}.{sh[i]}.{cu[i]},{simd[i]},{us[i,1]:.0f},{us[i,3]:.0f})" for i in last))
