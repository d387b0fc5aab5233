"""CPU tests (-m "not gpu") of the N > 1 path with world_size 2 over gloo: client sharding, the block broadcast
(the path's only exchange step) and the max-over-ranks timing reduce of bench.py.  The per-rank compute is done
by the oracle here (test stand-in for the HIP engine, which needs a GPU)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    import siggen
    from pyoracle import Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 6
    mine = bench.shard_clients(total, world, rank)
    taps = Oracle.lpf(1.0, bench.FS, bench.RATE // 2, bench.RATE)[1]  # 101 taps
    filters = {c: Oracle(bench.D, taps, bench.client_center_freq(c), bench.FS, bench.BLOCK_BYTES) for c in mine}
    recv = torch.empty(bench.BLOCK_BYTES, dtype=torch.uint8)
    results = {}
    for k in range(2):
        src = torch.from_numpy(siggen.xs_u8(77 + k, bench.BLOCK_BYTES)) if rank == 0 else None
        blk = bench.broadcast_block(dist, recv, src, rank)
        x = blk.numpy().copy()
        for c, f in filters.items():
            results[(k, c)] = f.process("cu8", x)
    tmax = bench.reduce_max_seconds(dist, torch, 1.0 + rank, "cpu")
    q.put((rank, mine, {k: v.tobytes() for k, v in results.items()}, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_broadcast_and_timing():
    import torch.multiprocessing as mp

    sys.path.insert(0, ROOT)
    import bench
    import siggen
    from pyoracle import Oracle

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = [q.get(timeout=180) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    got.sort()
    shards = [g[1] for g in got]
    assert sorted(shards[0] + shards[1]) == list(range(6)) and not set(shards[0]) & set(shards[1])
    assert all(abs(g[3] - 2.0) < 1e-12 for g in got)  # MAX over ranks of (1.0, 2.0)
    # every client's stream equals the single-process result on the same blocks
    taps = Oracle.lpf(1.0, bench.FS, bench.RATE // 2, bench.RATE)[1]
    merged = {}
    for g in got:
        merged.update(g[2])
    for c in range(6):
        f = Oracle(bench.D, taps, bench.client_center_freq(c), bench.FS, bench.BLOCK_BYTES)
        for k in range(2):
            want = f.process("cu8", siggen.xs_u8(77 + k, bench.BLOCK_BYTES))
            assert merged[(k, c)] == want.tobytes(), (k, c)


def test_shard_clients_balanced():
    sys.path.insert(0, ROOT)
    import bench

    for world in (1, 2, 4, 8):
        shards = [bench.shard_clients(1024 * world, world, r) for r in range(world)]
        assert all(len(s) == 1024 for s in shards)
        assert sorted(sum(shards, [])) == list(range(1024 * world))


def test_workload_accounting():
    sys.path.insert(0, ROOT)
    import bench

    assert bench.D == 42 and bench.S == 131072
    assert abs(bench.algorithmic_bytes_per_unit(42) - 2.190476) < 1e-5
    assert abs(bench.flops_per_unit(505, 42) - 96.33) < 0.01
    assert abs(bench.flops_per_unit(101, 42) - 19.38) < 0.01
