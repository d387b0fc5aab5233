#!/usr/bin/env python3
"""tools/fwdrole_trace.py <dump> -- timeline of a two-half mix launch that carries the forward role (xl_fwd_role.h), from the dump a
-DXL_TUNING library (tools/experiments/build_tune.sh) writes with XL_EXP_POLY_TRACE=<file>: role workgroups (slots 6000..: start, end,
samples arrived, pass) and mix waves (slots 0..: start, end, -, first pass over); 100 MHz clock."""
import sys
import numpy as np
h = np.fromfile(sys.argv[1], dtype=np.uint64)
w = h[4096:4096 + 4 * 7168].reshape(7168, 4).astype(np.int64)
mix, role = w[:6000], w[6000:]
mix, role = mix[mix[:, 0] > 0], role[role[:, 0] > 0]
t0 = min(mix[:, 0].min(), role[:, 0].min()) if len(role) else mix[:, 0].min()
us = lambda c: (c - t0) * 0.01
q = lambda x: "min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % (x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max())
if len(role):
    print(f"role workgroups {len(role)}: start {q(us(role[:, 0]))}")
    print(f"  samples arrived after {q((role[:, 2] - role[:, 0]) * 0.01)}; transforms + stores + publish {q((role[:, 1] - role[:, 2]) * 0.01)}; end {q(us(role[:, 1]))}")
    for p in sorted(set(role[:, 3])):
        r = role[role[:, 3] == p]
        print(f"  pass {p}: {len(r)} workgroups, start med {np.median(us(r[:, 0])):.1f}, complete at {us(r[:, 1]).max():.1f}")
print(f"mix waves {len(mix)}: start {q(us(mix[:, 0]))}")
print(f"  first pass over {q(us(mix[:, 3]))} (takes {q((mix[:, 3] - mix[:, 0]) * 0.01)}); end {q(us(mix[:, 1]))}; launch span {us(mix[:, 1]).max():.1f} us")
for t in (1, 3, 5, 8, 10, 15, 20, 25, 30, 40, 50, 60, 70):
    print(f"  t={t:2d} us: role alive {int(((us(role[:, 0]) <= t) & (us(role[:, 1]) > t)).sum()) if len(role) else 0}  mix waves alive {int(((us(mix[:, 0]) <= t) & (us(mix[:, 1]) > t)).sum())} in first pass {int(((us(mix[:, 0]) <= t) & (us(mix[:, 3]) > t)).sum())}")
