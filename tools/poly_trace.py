#!/usr/bin/env python3
"""tools/poly_trace.py <dump> <nco_blocks> <work_blocks> -- timeline of one xlp_mix_kernel launch from the dump written
with XL_EXP_POLY_TRACE=<file> (per NCO wave: start / phase loaded / recurrence done / end / placement; per work
workgroup: start / end / placement)."""
import sys
import numpy as np

h = np.fromfile(sys.argv[1], dtype=np.uint64)
nn, nw = int(sys.argv[2]), int(sys.argv[3])


def place(v):
    hw, xcc = int(v) & 0xFFFFFFFF, (int(v) >> 32) & 7
    return (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)  # xcc, se, sh, cu, simd


w = h[4096:4096 + 4 * nw].reshape(nw, 4)
t0 = int(w[:, 0][w[:, 0] > 0].min())
ws = (w[:, 0].astype(np.int64) - t0) * 0.01
we = (w[:, 1].astype(np.int64) - t0) * 0.01
wp = [place(v) for v in w[:, 2]]
print(f"work workgroups: {nw}; start min {ws.min():.1f} med {np.median(ws):.1f} max {ws.max():.1f}; end min {we.min():.1f} p10 {np.percentile(we,10):.1f} "
      f"med {np.median(we):.1f} p90 {np.percentile(we,90):.1f} max {we.max():.1f} us")
dur = we - ws
print(f"work workgroup durations: min {dur.min():.1f} p10 {np.percentile(dur,10):.1f} med {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} max {dur.max():.1f} us; "
      f"started after 2 us: {(ws > 2).sum()} (their durations med {np.median(dur[ws > 2]) if (ws > 2).any() else 0:.1f})")
nco = h[8:8 + 8 * nn].reshape(nn, 8)
simd_of = {}
for i in range(nn):
    p = place(nco[i, 5])
    simd_of[p] = i
    st = [(int(nco[i, k]) - t0) * 0.01 for k in range(4)]
    same = [j for j in range(nw) if wp[j] == p]
    samecu = [j for j in range(nw) if wp[j][:4] == p[:4] and wp[j] != p]
    print(f"nco wave {i:2d} at xcc{p[0]} se{p[1]} sh{p[2]} cu{p[3]:2d} simd{p[4]}: start {st[0]:5.1f} loaded {st[1]:5.1f} chain done {st[2]:5.1f} end {st[3]:5.1f} | "
          f"work waves on its SIMD: {len(same)} end {[round(float(we[j]),1) for j in same]} | same CU other SIMDs: {len(samecu)} end med "
          f"{np.median([we[j] for j in samecu]) if samecu else 0:.1f}")
vict = [j for j in range(nw) if wp[j] in simd_of]
rest = [j for j in range(nw) if wp[j] not in simd_of]
print(f"victims (share a SIMD with an NCO wave): {len(vict)} end med {np.median(we[vict]):.1f} max {we[vict].max():.1f};  others: end med {np.median(we[rest]):.1f} p99 {np.percentile(we[rest],99):.1f} max {we[rest].max():.1f}")
late = np.argsort(-we)[:12]
print("latest:", [(int(j), round(float(we[j]), 1), wp[j], wp[j] in simd_of) for j in late])
