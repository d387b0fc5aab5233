#!/usr/bin/env python3
"""tools/replan_cost.py -- GPU: what a client joining / leaving costs the engine (re-plan), 1024 clients, 8 blocks per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import siggen, sdr_server_amd as xl

FS, D, BLOCK, G = 2016000, 42, 262144, 8
taps = xl.create_low_pass_filter(1.0, FS, 24000, 9600)[1]
for n in (128, 1024, 4096):
    eng = xl.BatchEngine(FS, "cu8", BLOCK, group_blocks=G)
    for c in range(n):
        eng.add_client(D, taps, -984000 + 1920 * (c % 1024) + 240 * (c // 1024))
    data = torch.from_numpy(siggen.xs_u8(99, G * BLOCK)).cuda()

    def call():
        eng.process_device_group(data.data_ptr(), BLOCK, G, "optimized", "engine")

    for _ in range(4):
        call()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        call()
    eng.sync()
    steady = (time.perf_counter() - t0) / 10
    res = []
    for trial in range(3):
        t0 = time.perf_counter()
        cid = eng.add_client(D, taps, 12345 + trial)
        t1 = time.perf_counter()
        call()                  # plan with the newcomer (its own direct class)
        t1b = time.perf_counter()
        eng.sync()
        t2 = time.perf_counter()
        print(f"   trial {trial}: call after join: host {1e3*(t1b-t1):.3f} ms + sync {1e3*(t2-t1b):.3f} ms", file=sys.stderr)
        call(); eng.sync()      # the newcomer is mature: re-plan, it joins the polyphase class
        t3 = time.perf_counter()
        call(); eng.sync()
        t4 = time.perf_counter()
        eng.remove_client(cid)
        call(); eng.sync()
        t5 = time.perf_counter()
        res.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
    eng.close()
    f = lambda i: 1e3 * min(r[i] for r in res)
    print(f"clients {n:5d}: steady call {steady*1e3:7.3f} ms | add_client {f(0):6.3f} ms, first call after join {f(1):7.3f} ms, call after maturing {f(2):7.3f} ms, "
          f"next call {f(3):7.3f} ms, call after leave {f(4):7.3f} ms")
