#!/usr/bin/env python3
"""tools/feed_nccl_selftest.py -- bench.py's multi-GPU super-block feed over the REAL RCCL backend on a one-GPU box: a
world-size-1 process group (the broadcast is then trivial, but every call -- init with device_id, broadcast issued on the
side stream, event ordering, buffer reuse -- is the one the N > 1 run makes), the feeder driven as rank 0 of 2.
Checks that every client's output equals the directly-fed run bit for bit.  Run: python tools/feed_nccl_selftest.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
import sdr_server_amd as xl  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev_groups = [torch.from_numpy(bench.make_group(g)).cuda() for g in range(bench.NSRC_GROUPS)]
taps = xl.create_low_pass_filter(1.0, bench.FS, 24000, 9600)[1]


def run(world):
    eng = xl.BatchEngine(bench.FS, "cu8", bench.BLOCK_BYTES, group_blocks=bench.GROUP)
    for c in range(256):
        eng.add_client(bench.D, taps, -984000 + 1920 * c)
    feeder = bench.GroupFeeder(torch, dist, 0, world, dev_groups)
    stream = torch.cuda.current_stream()
    outs = []
    for k in range(12):
        ptr = feeder.get(k, stream)
        eng.process_device_group(ptr, bench.BLOCK_BYTES, bench.GROUP, "optimized", stream.cuda_stream)
        feeder.consumed(k, stream)
        if k % 5 == 3:
            torch.cuda.synchronize()
            eng.fetch()
            outs.append([eng.output(c).copy() for c in (0, 100, 255)])
    torch.cuda.synchronize()
    eng.close()
    return outs


a, b = run(1), run(2)
ok = all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for oa, ob in zip(a, b) for x, y in zip(oa, ob))
print("feed over RCCL (world-size-1 group, feeder as rank 0 of 2):", "outputs identical to the direct feed" if ok else "MISMATCH")
dist.destroy_process_group()
sys.exit(0 if ok else 1)
