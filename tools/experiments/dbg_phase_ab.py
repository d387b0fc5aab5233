"""Two engines on the same stream of one-block optimized calls, one with the FMA mix, one with the matrix-core mix: the committed NCO
phases of all clients must agree bit for bit after every call (the recurrence does not depend on the mix).  Prints the mismatches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import siggen, sdr_server_amd as xl
FS = 2016000
N = int(os.environ.get("AB_CLIENTS", "1024")); CALLS = int(os.environ.get("AB_CALLS", "150")); G = int(os.environ.get("AB_G", "1"))
code, t48 = xl.create_low_pass_filter(1.0, FS, 24000, 9600)
engs = []
for mix in (0, 1):
    os.environ["XL_EXP_MIX"] = str(mix)
    e = xl.BatchEngine(FS, "cu8", 262144, group_blocks=G)
    ids = [] if os.environ.get("AB_STAGGER") else [e.add_client(42, t48, -984000 + 1920 * c) for c in range(N)]
    engs.append((e, ids))
print(engs[1][0].describe())
bad_total = {}
for k in range(CALLS):
    x = siggen.xs_u8(7000 + k, G * 262144)
    if os.environ.get("AB_STAGGER") and k < 21:
        for e, ids in engs:
            for c in range(49 * k, min(N, 49 * (k + 1))): ids.append(e.add_client(42, t48, -984000 + 1920 * c))
    ph = []
    for e, ids in engs:
        if G == 1: e.process_host(x, "optimized")
        else: e.process_host_group(x, G, "optimized")
        e.sync()
        ph.append(np.array([e.phase(i) for i in ids], dtype=np.float32))
    if ph[0].shape != ph[1].shape or len(ph[0]) == 0: continue
    d = np.flatnonzero((ph[0].view(np.uint32) != ph[1].view(np.uint32)).any(axis=1))
    new = [int(c) for c in d if int(c) not in bad_total]
    for c in new: bad_total[c] = k
    if new: print("call", k, "new bad clients", new[:40], "lanes", sorted(set(c % 64 for c in new)))
print("total bad", len(bad_total), "of", N, "after", CALLS, "calls")
