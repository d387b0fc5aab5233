#!/usr/bin/env python3
"""tools/feed_overhead.py -- GPU (one device): what bench.py's multi-GPU block feed costs per block, measured with a
stand-in for torch.distributed whose broadcast is a device-to-device copy on the communication stream (rank 0's view:
copy into the receive buffer + "broadcast").  Compares a resident block (N = 1 path) with the two-buffer event-ordered
feed (N > 1 path) for the bench workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench, sdr_server_amd as xl


class FakeDist:
    def broadcast(self, t, src=0):
        return None


def run(world, steps=300, sb=1):
    code, taps = xl.create_low_pass_filter(1.0, bench.FS, bench.RATE // 2, bench.RATE // 5)
    eng = xl.BatchEngine(bench.FS, "cu8", bench.BLOCK_BYTES)
    for c in range(1024):
        eng.add_client(bench.D, taps, bench.client_center_freq(c))
    dev = [torch.from_numpy(b).cuda() for b in bench.make_blocks(8, 1)]
    feeder = bench.BlockFeeder(torch, FakeDist(), 0, world, dev, group=sb)
    stream = torch.cuda.current_stream()

    def step(k):
        ptr = feeder.get(k, stream)
        eng.process_device(ptr, bench.BLOCK_BYTES, "optimized", stream.cuda_stream)
        feeder.consumed(k, stream)

    for k in range(20):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(20 + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    eng.close()
    return dt


for w, g in ((1, 1), (2, 1), (2, 8)):
    bench.FEED_GROUP = g
    dt = run(w, sb=g)
    print(f"feed as for N={w}, {g} block(s) per broadcast: {dt*1e6:.1f} us per block  ({1024*bench.S/dt/1e6:.0f} Msps per GPU)")
