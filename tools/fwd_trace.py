#!/usr/bin/env python3
"""tools/fwd_trace.py <dump> <workgroups> -- phases of the polyphase forward launch from the dump written with
XL_EXP_POLY_TRACE=<file> XL_EXP_POLY_TRACE_FWD=1 by a -DXL_TUNING build (per workgroup, thread 0: start / samples and
twiddles arrived / transforms in LDS and image stores issued / stores acknowledged); 100 MHz clock."""
import sys
import numpy as np
h = np.fromfile(sys.argv[1], dtype=np.uint64)
nw = int(sys.argv[2])
w = h[4096:4096 + 4 * nw].reshape(nw, 4).astype(np.int64)
w = w[w[:, 0] > 0]
t0 = w[:, 0].min()
us = lambda c: (c - t0) * 0.01
q = lambda x: "min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % (x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max())
print(f"workgroups traced {len(w)}; launch span {us(w[:, 1]).max():.1f} us")
print("start        :", q(us(w[:, 0])))
print("loaded       :", q(us(w[:, 2])), "| load phase", q((w[:, 2] - w[:, 0]) * 0.01))
print("stores issued:", q(us(w[:, 3])), "| transform + transpose", q((w[:, 3] - w[:, 2]) * 0.01))
print("end          :", q(us(w[:, 1])), "| store acknowledge", q((w[:, 1] - w[:, 3]) * 0.01))
for t in (1, 2, 4, 6, 8, 10, 12, 14, 16, 18):
    print(f"  t={t:2d} us: alive {int(((us(w[:, 0]) <= t) & (us(w[:, 1]) > t)).sum())}")
