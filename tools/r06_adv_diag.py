#!/usr/bin/env python3
"""diagnostic: where the adversarial cf32 call's errors sit (per block, worst client, output index -> segment)"""
import os, sys
os.environ.setdefault("XL_TESTING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import siggen, sdr_server_amd as xl
from pyoracle import Oracle
import test_batch_gpu as T

nsamp = 131072
for shape in sys.argv[1:] or ["config5_d100"]:
    if shape == "config5_d100":
        fs, D, taps = 10000000, 100, siggen.hamming_sinc(257, 0.004)
    else:
        fs, D, taps = 2016000, 42, T.lpf(2016000, 24000, 9600)
    blocks = T._cf32_adversarial_blocks(nsamp)
    ncl = 40
    fcs = [int(-0.45 * fs + (0.9 * fs / ncl) * c) for c in range(ncl)]
    orc = [Oracle(D, taps, fc, fs, 8 * nsamp) for fc in fcs]
    warm = (siggen.xs_s16(17, 2 * nsamp).astype(np.float32) / np.float32(32768)).astype(np.float32)
    [o.process("cf32", warm) for o in orc]
    want = [[o.process("cf32", xb) for xb in blocks] for o in orc]
    for mix in (0, 3):
        eng = xl.BatchEngine(fs, "cf32", 2 * nsamp, group_blocks=8)
        if mix: eng.set_option("mix_kernel", mix)
        ids = [eng.add_client(D, taps, fc) for fc in fcs]
        eng.process_host(warm, "optimized"); eng.fetch()
        eng.process_host_group(np.concatenate(blocks), 8, "optimized"); eng.fetch()
        plan = eng.describe()
        V = int(plan.split(" V")[1].split()[0])
        print(shape, "mix", mix, plan.split("|")[2].strip())
        for g in range(8):
            worst = (0, -1, -1)
            for c in range(ncl):
                got = eng.output(ids[c]); lens = [eng.output_len_block(ids[c], k) for k in range(8)]
                off = sum(lens[:g]); gb = got[off:off + lens[g]]; wb = want[c][g]
                den = max(np.abs(wb).max(), 1e-300)
                d = np.abs(gb.astype(np.complex128) - wb)
                i = int(d.argmax())
                if d[i] / den > worst[0]: worst = (d[i] / den, c, i, off + i, den, float(np.abs(wb[i])))
            print("  block", g, "worst rel %.3g client %d at output %d of the block (call output %d, segment %d, pos %d of V=%d) block max|y| %.3g |y| there %.3g" % (
                worst[0], worst[1], worst[2], worst[3], worst[3] // V, worst[3] % V, V, worst[4], worst[5]))
        eng.close()
