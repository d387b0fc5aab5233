#!/usr/bin/env python3
"""tools/experiments/dbg_soak.py -- who corrupts NCO phases in one-block calls with the recurrence inside the launches?
4096 clients, N one-block calls; committed phases of all clients at checkpoints.  Configurations:
  ref   nco_side_stream = 1 (the chain kernel), matrix-core mix          -- run alone
  fma   nco_side_stream = 0, packed-FMA mix (role in forward, mix, inverse) -- run alone
  mfma  nco_side_stream = 0, matrix-core mix (role in forward, inverse)     -- run alone
  then fma + mfma TOGETHER in one process (interleaved calls, as the soak test does), and ref + ref together.
Prints, per configuration, the first checkpoint at which its phases differ from `ref` and which clients."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import siggen  # noqa: E402
import sdr_server_amd as xl  # noqa: E402

FS, NB = 2016000, 262144
N = int(os.environ.get("SOAK_CALLS", "1500"))
NC = int(os.environ.get("SOAK_CLIENTS", "4096"))
EVERY = 100
code, taps = xl.create_low_pass_filter(1.0, FS, 24000, 9600)
blocks = [siggen.xs_u8(7300 + k, NB) for k in range(4)]


def make(mix, side):
    os.environ["XL_EXP_MIX"] = str(mix)
    e = xl.BatchEngine(FS, "cu8", NB)
    e.set_option("nco_side_stream", side)
    ids = [e.add_client(42, taps, -984000 + 480 * c) for c in range(NC)]
    return e, ids


def phases(e, ids):
    e.sync()
    return np.array([e.phase(i) for i in ids], dtype=np.float32).view(np.uint32)


def run(engs):
    out = [[] for _ in engs]
    for k in range(N):
        for e, _ in engs:
            e.process_host(blocks[k % 4], "optimized")
        if k % EVERY == EVERY - 1:
            for j, (e, ids) in enumerate(engs):
                out[j].append(phases(e, ids))
    for e, _ in engs:
        e.close()
    return out


def report(name, got, ref):
    for cp, (g, r) in enumerate(zip(got, ref)):
        bad = np.flatnonzero((g != r).any(axis=1))
        if len(bad):
            print(f"{name:28s} FIRST MISMATCH at call {(cp + 1) * EVERY}: {len(bad)} clients, e.g. {bad[:24].tolist()}  (lanes {sorted(set((bad % 64).tolist()))[:8]}..)", flush=True)
            return
    print(f"{name:28s} equal to ref at all {len(got)} checkpoints", flush=True)


ref = run([make(1, 1)])[0]
if os.environ.get("SOAK_ONLY_PAIR"):  # (variant builds: only the configuration that fails)
    two = run([make(0, 0), make(1, 0)])
    report("fma  (with mfma beside it)", two[0], ref)
    report("mfma (with fma beside it)", two[1], ref)
    sys.exit(0)
report("ref again (alone)", run([make(1, 1)])[0], ref)
report("fma alone (role x3)", run([make(0, 0)])[0], ref)
report("mfma alone (role x2)", run([make(1, 0)])[0], ref)
two = run([make(0, 0), make(1, 0)])
report("fma  (with mfma beside it)", two[0], ref)
report("mfma (with fma beside it)", two[1], ref)
two = run([make(1, 1), make(1, 1)])
report("ref A (two side-stream engines)", two[0], ref)
report("ref B (two side-stream engines)", two[1], ref)
