#!/usr/bin/env python3
"""tools/experiments/dbg_soak2.py -- do WORK waves (packed-FMA mix / inverse launches) miscompute next to another engine's
matrix-core launches, or only the in-launch NCO role?  Both engines tabulate their phases on the side stream (exclusive SIMDs:
immune), so any difference is in the filtering itself.  The packed-FMA engine's outputs (all clients, hashed per client) at
checkpoints, run ALONE and then beside a matrix-core engine: bitwise equal or not."""
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import siggen  # noqa: E402
import sdr_server_amd as xl  # noqa: E402

FS, NB = 2016000, 262144
N = int(os.environ.get("SOAK_CALLS", "600"))
NC = int(os.environ.get("SOAK_CLIENTS", "4096"))
EVERY = 50
code, taps = xl.create_low_pass_filter(1.0, FS, 24000, 9600)
blocks = [siggen.xs_u8(7300 + k, NB) for k in range(4)]


def make(mix):
    os.environ["XL_EXP_MIX"] = str(mix)
    e = xl.BatchEngine(FS, "cu8", NB)
    e.set_option("nco_side_stream", 1)
    ids = [e.add_client(42, taps, -984000 + 480 * c) for c in range(NC)]
    return e, ids


def outputs(e, ids):
    e.fetch()
    return np.array([zlib.crc32(e.output(i).tobytes()) for i in ids], dtype=np.uint32)


def run(engs, watch):
    out = []
    for k in range(N):
        for e, _ in engs:
            e.process_host(blocks[k % 4], "optimized")
        if k % EVERY == EVERY - 1:
            out.append(outputs(*engs[watch]))
    for e, _ in engs:
        e.close()
    return out


for mix, name in ((0, "packed-FMA engine"), (1, "matrix-core engine")):
    alone = run([make(mix)], 0)
    again = run([make(mix)], 0)
    beside = run([make(mix), make(1)], 0)
    for label, got in (("alone, second run", again), ("beside a matrix-core engine", beside)):
        bad = [(cp, int((g != a).sum())) for cp, (g, a) in enumerate(zip(got, alone)) if (g != a).any()]
        print(f"{name:20s} {label:30s}: " + ("outputs bit-equal at all %d checkpoints" % len(got) if not bad else
              "DIFFERENT outputs at %d of %d checkpoints, first: call %d, %d clients" % (len(bad), len(got), (bad[0][0] + 1) * EVERY, bad[0][1])), flush=True)
