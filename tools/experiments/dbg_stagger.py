import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import siggen, sdr_server_amd as xl
from pyoracle import Oracle
FS = 2016000
code, t48 = xl.create_low_pass_filter(1.0, FS, 24000, 9600)
eng = xl.BatchEngine(FS, "cu8", 262144)
fcs = [-984000 + 1920 * c for c in range(1024)]
per = [49] * 20 + [44]
oracles = {}
nxt = 0
def run(x, variant):
    eng.process_host(x, variant); eng.fetch()
    bad = []
    for cid, o in oracles.items():
        want = o.process("cu8", x); got = eng.output(cid)
        d = np.flatnonzero(got.view(np.uint64) != want.view(np.uint64)) if variant == "native" else np.flatnonzero(np.abs(got - want) > 1e-5 * np.abs(want).max())
        if len(d): bad.append((cid, int(d[0]), int(d[-1]), len(d), len(want)))
    return bad
for k in range(24):
    if k < 21:
        for _ in range(per[k]):
            cid = eng.add_client(42, t48, fcs[nxt])
            if nxt % 16 == 0 or nxt in (5, 47, 48, 49, 95, 96): oracles[cid] = Oracle(42, t48, fcs[nxt], FS, 262144)
            nxt += 1
    b = run(siggen.xs_u8(4700 + k, 262144), "optimized")
    if b: print("k", k, "optimized bad", b[:6])
print(eng.describe())
print("native bad:", run(siggen.xs_u8(4790, 262144), "native"))
print("native again bad:", run(siggen.xs_u8(4792, 262144), "native"))
print("opt bad:", run(siggen.xs_u8(4791, 100002), "optimized"))
