#!/usr/bin/env python3
"""tools/inv_trace.py <dump> <work_workgroups> -- phases of the polyphase inverse launch from the dump written with
XL_EXP_POLY_TRACE=<file> XL_EXP_POLY_TRACE_INV=1 (per workgroup, first wave: start / tile in LDS / transforms done / end)."""
import sys
import numpy as np
h = np.fromfile(sys.argv[1], dtype=np.uint64)
nw = int(sys.argv[2])
w = h[4096:4096 + 4 * nw].reshape(nw, 4).astype(np.int64)
w = w[w[:, 0] > 0]
t0 = w[:, 0].min()
us = lambda c: (c - t0) * 0.01
q = lambda x: "min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % (x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max())
print(f"workgroups {len(w)}")
print("start       :", q(us(w[:, 0])))
print("tile loaded :", q(us(w[:, 2])), "| load phase", q((w[:, 2] - w[:, 0]) * 0.01))
print("transformed :", q(us(w[:, 3])), "| transform phase", q((w[:, 3] - w[:, 2]) * 0.01))
print("end         :", q(us(w[:, 1])), "| store phase", q((w[:, 1] - w[:, 3]) * 0.01))
