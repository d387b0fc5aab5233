#!/usr/bin/env python3
"""bench.py -- throughput of the xlating-FIR hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]

N > 1 without a torch.distributed environment re-launches itself under `python -m torch.distributed.run` (one rank per
GPU, rendezvous on 127.0.0.1); inside such a launch (RANK / WORLD_SIZE set, e.g. by the driver) it just runs its rank.

Workload (config.workload) = BASELINE.json configs[3] / the 1-GPU target: 1024 concurrent 48 kHz clients PER GPU
(`--scaling weak`, the default: N x 1024 clients, client c -> GPU c mod N; `--scaling strong`: 1024 IN TOTAL = configs[3]
read literally -- at N > 1 the line carries the other one as variants["strong scaling ..."]; one GPU runs 1024 clients within
10 % of the floor the reference's phase recurrence sets for ANY client count, so a fixed total does not scale and VERDICT r2
asked for the weak curve as the multi-GPU headline; at N = 1 the two are the same run) off one 2.016 Msps cu8
stream, server-default 262144-byte blocks (131072 complex samples), D = 42, low-pass designed with the server default
lpf_cutoff_rate=5 -> 505 taps (the BASELINE "~84-tap" figure is not reachable with the reference's designer,
SURVEY D3).  The engine is driven as sdr_callback would drive it with a super-block: calls
of 8 consecutive blocks (xlating_batch_process_device_group) -- results identical to 8 successive process calls.

A "step" = one pass of the hot path over one batch of synthetic input = BLOCKS_PER_STEP consecutive blocks (240 calls of
8 blocks, 252 M samples of the stream) for every client of every GPU.  The timed region (exactly `--steps` steps between
barrier + synchronize on both sides, max over ranks) is repeated REPEATS = 3 times; `value` / `ms_per_step` are the
MEDIAN repeat, all three are printed ("repeats").  20 steps ~ 1.1 s per repeat at 1024 clients.
    N > 1: rank 0's blocks are broadcast over RCCL/xGMI, 8 blocks per broadcast, on a side stream (the path's only
    exchange step), then each rank runs its own clients on them -- no further communication.
Inputs are synthetic (xorshift bytes), already resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0): value = input IQ Msamples/s summed over all clients and GPUs.
"roofline":     PHYSICAL: "traffic" = HBM bytes one call's launches move by the PMC counters (FETCH_SIZE / WRITE_SIZE in
                separate rocprofv3 passes over `bench.py --replay-calls`, run as subprocesses right after the timed region
                -- "traffic_source" says "measured in this run", or names the committed file it fell back to), "achieved" =
                traffic / the mean duration of a call's launches (HIP events on the launch stream inside the timed region),
                "frac" = achieved / 8 TB/s: <= 1 by construction.  "per_kernel": duration, bytes, HBM and FP32 fractions and
                the ceiling that binds each launch.  The per-client-read MODEL of SURVEY 8(d) (2 B in + 8/D B out per
                (client, sample): every client "reads" the block the engine reads once) is kept as "model_frac" -- it is
                not traffic and can exceed 1.
"parity_spot":  after the timed region EVERY client of this GPU is fetched and compared with a population of oracle filters
                run on the host cores (oracle/population.c; stream state fast-forwarded over the run: phase recurrence
                only, then one real super-block to load the history, the next one compared).
"native":       the same workload with the reference's default arithmetic (cpu_optimization NATIVE_CF32,
                src/config.c:252-264): bit-exact scalar order, direct FIR kernel.
"variants":     context for the headline, each a short run of its own on this box: other filter lengths / call granularities / client
                counts (2048 and 4096: where the launches, not the recurrence, bound the call -- the 2048 one with its own counters
                and EVERY client checked); "polyphase, float32 matrix-core mix": the all-float32 arithmetic (option mix_kernel = 3),
                every client checked; "inverse launch A/B in this process": the size rule's inverse kernel and its alternate alternating on this box;
                "config 5 ...": BASELINE configs[4] (cf32 10 Msps, D = 100, 257 taps, 1024 clients) with its own roofline block
                (counter bytes, shared-read algorithmic fraction, FP32 / matrix-f32 fractions) and every client checked;
                "host-delivered outputs": process_host + fetch per call (PCIe-inclusive; never `value`).
"roofline.frac_algorithmic_shared": SURVEY 8(d)'s shared-read minimum (2/N B in + 8/D B out per (client, sample)) over the launches' own
                time / 8 TB/s -- the USEFUL fraction, next to the counter fraction `frac`.
"cpu_baseline": the reference itself (oracle/_ref, unmodified sources, -O3 -ffast-math AVX2) -- or the repo's CPU
                restatement when that build is absent -- timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

FS = 2016000
RATE = 48000
D = FS // RATE
BLOCK_BYTES = 262144            # server default buffer_size (src/resources/config.conf:13)
S = BLOCK_BYTES // 2            # complex samples per block
GROUP = 8                       # blocks per engine call and per RCCL broadcast (8-block super-block)
BLOCKS_PER_STEP = 1920          # blocks per bench step (240 calls): 20 steps ~ 1.1 s of GPU time at 1024 clients
REPEATS = 3                     # the timed region is repeated; value = the median repeat
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: FP32 vector == FP32 MFMA peak
MFMA_F16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense BF16 / F16 MFMA peak
TIMING_STRIDE = 5               # HIP events bracket every 5th call of the timed region (an event pair costs ~6 us of stream time; odd, so that both calls of a chain launch's pair are sampled)
PMC_REPLAY_CALLS = 12           # calls of the counter passes (rocprofv3 serialises the dispatches; the first 3 per kernel are dropped)
# BASELINE configs[4] as SURVEY D4 / 8(d) define it: Airspy-style cf32 input at 10 Msps, D = 100 (-> 100 kHz: 96 kHz is no integer
# decimation of 10 MHz and the server would reject it), 257 explicit Hamming-sinc taps, 131072 complex samples per block
C5_FS, C5_D, C5_TAPS, C5_CUTOFF = 10000000, 100, 257, 0.004


def shard_clients(total_clients, world_size, rank):
    """Client c -> GPU (c mod G); returns this rank's global client indices."""
    return [c for c in range(total_clients) if c % world_size == rank]


def client_center_freq(c):
    """Distinct in-band offsets: 1920 Hz raster across +-984 kHz, shifted by 240 Hz per 1024 clients."""
    return -984000 + 1920 * (c % 1024) + 240 * ((c // 1024) % 8)


def algorithmic_bytes_per_unit(decim):
    """per (client, complex input sample): 2 B cu8 in + 8/D B cf32 out."""
    return 2.0 + 8.0 / decim


def flops_per_unit(ntaps, decim):
    """8 flop per complex MAC, T/D MACs per input sample, + 6/D for the NCO rotate."""
    return 8.0 * ntaps / decim + 6.0 / decim


def broadcast_block(dist, recv, src_block, rank):
    """The path's only exchange step: rank 0's raw IQ blocks -> every rank (RCCL over xGMI on GPUs, gloo in the
    CPU tests).  `recv` is each rank's receive buffer; returns the tensor holding the blocks on this rank."""
    if rank == 0:
        recv.copy_(src_block, non_blocking=True)
    dist.broadcast(recv, src=0)
    return recv


def reduce_max_seconds(dist, torch, dt, device):
    """bench contract: the step time is the MAX over ranks."""
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_seconds(dist, torch, dt, device, world):
    """every rank's time of one timed repeat (rank order), next to the max the contract asks for"""
    t = torch.zeros(world, dtype=torch.float64, device=device)
    t[dist.get_rank()] = dt
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [round(float(v), 6) for v in t.tolist()]


NSRC_GROUPS = 4  # distinct 8-block super-blocks the synthetic stream cycles through


def make_group(g, seed=0x5DEECE66D):
    """Super-block g of the synthetic stream: GROUP blocks back to back (uint8)."""
    import siggen

    return siggen.xs_u8(seed + g, GROUP * BLOCK_BYTES)


def self_launch(args, argv):
    """`python bench.py --gpus N` outside a torch.distributed environment: start N ranks of this script on this node
    (what the driver's own `python -m torch.distributed.run ... bench.py` command does) and pass their output through."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------- engines
class PlumbingEngine:
    """CPU stand-in for the HIP engine, used ONLY by `--plumbing-test` (tests/test_bench_launch.py: the launcher,
    sharding, feed and reduce code paths on a GPU-less box).  It filters nothing and its numbers are never reported as
    a measurement ("data": "cpu-plumbing-test")."""

    def __init__(self):
        self.n = 0
        self.calls = 0

    def add_client(self, *a):
        self.n += 1
        return self.n - 1

    def process_device_group(self, ptr, input_len, nblocks, mode, stream=0):
        self.calls += 1

    def describe(self):
        return f"clients {self.n} classes 1 | direct: none | polyphase: none"

    def timing(self, *a):
        pass

    timing_stride = sync = close = timing

    def timing_read(self, reset=True):
        return 0, 0.0, 0.0

    def output_len(self, cid):
        return 0


class GroupFeeder:
    """Delivers super-block k (GROUP blocks) to this rank: N = 1 -> a resident device buffer; N > 1 -> rank 0's
    super-blocks broadcast over RCCL/xGMI into one of two alternating receive buffers ON A SEPARATE STREAM, so that
    the broadcast of super-block k+1 overlaps the filtering of k (events order buffer reuse: the broadcast into a
    buffer waits until the launches that read it two calls ago have been passed by the compute stream)."""

    def __init__(self, torch, dist, rank, world, dev_groups, cuda=True):
        self.torch, self.dist, self.rank, self.world, self.groups, self.cuda = torch, dist, rank, world, dev_groups, cuda
        self.issued = 0
        if world > 1:
            dev = "cuda" if cuda else "cpu"
            self.recv = [torch.empty(GROUP * BLOCK_BYTES, dtype=torch.uint8, device=dev) for _ in range(2)]
            if cuda:
                self.comm = torch.cuda.Stream()
                self.ready = [torch.cuda.Event() for _ in range(2)]
                self.free = [torch.cuda.Event() for _ in range(2)]
            self.free_valid = [False, False]

    def _issue(self, g):
        torch = self.torch
        i = g % 2
        if not self.cuda:
            broadcast_block(self.dist, self.recv[i], self.groups[g % len(self.groups)] if self.rank == 0 else None, self.rank)
            self.issued = g + 1
            return
        if self.free_valid[i]:
            self.comm.wait_event(self.free[i])
        with torch.cuda.stream(self.comm):
            broadcast_block(self.dist, self.recv[i], self.groups[g % len(self.groups)] if self.rank == 0 else None, self.rank)
            self.ready[i].record(self.comm)
        self.issued = g + 1

    def get(self, k, stream):
        """Device pointer of super-block k, valid on `stream`; call consumed(k, stream) after enqueuing its consumer."""
        if self.world == 1:
            return self.groups[k % len(self.groups)].data_ptr()
        while self.issued <= k + 1:  # keep one broadcast in flight ahead of the consumer
            self._issue(self.issued)
        if self.cuda:
            stream.wait_event(self.ready[k % 2])
        return self.recv[k % 2].data_ptr()

    def consumed(self, k, stream):
        if self.world > 1 and self.cuda:
            self.free[k % 2].record(stream)
            self.free_valid[k % 2] = True


class TorchHost:
    """Feed through torch.distributed (GroupFeeder) + one BatchEngine on torch's current stream.  The fallback of the C
    host below, the path of `--feed torch`, and (with the no-op engine) of the CPU plumbing test."""

    name = "torch.distributed broadcast (python feeder) + xlating_batch"

    def __init__(self, ctx, group):
        xl, torch = ctx["xl"], ctx["torch"]
        self.ctx, self.cuda = ctx, ctx["cuda"]
        if self.cuda:
            self.eng = xl.BatchEngine(FS, "cu8", BLOCK_BYTES, device=torch.cuda.current_device(), group_blocks=group)
            self.stream = torch.cuda.current_stream()
            self.sptr = self.stream.cuda_stream
        else:
            self.eng, self.stream, self.sptr = PlumbingEngine(), None, 0
        self.feeder = GroupFeeder(torch, ctx["dist"], ctx["rank"], ctx["world"], ctx["dev_groups"], self.cuda)
        self.k = 0

    def add_client(self, c, taps):
        return self.eng.add_client(D, taps, client_center_freq(c))

    def call(self, mode, nblocks, first_block=0, count=None, advance=True):
        """Super-block k of the stream in calls of `nblocks` blocks (count: only that many blocks, from first_block;
        advance=False: more blocks of the same super-block follow)."""
        ptr = self.feeder.get(self.k, self.stream)
        last = GROUP if count is None else first_block + count
        for j in range(first_block, last, nblocks):
            self.eng.process_device_group(ptr + j * BLOCK_BYTES, BLOCK_BYTES, nblocks, mode, self.sptr)
        if advance:
            self.feeder.consumed(self.k, self.stream)
            self.k += 1

    def sync(self):
        if self.cuda:
            self.ctx["torch"].cuda.synchronize()

    def close(self):
        self.eng.close()


class CHost:
    """include/xlating_multi.h through ctypes: the engines, the RCCL broadcast of every super-block on a communication
    stream and the event plumbing all live in the C library (north_star: host code stays in C).  torch.distributed is
    used by bench.py only to hand rank 0's RCCL id to the other ranks and for the barrier / max-over-ranks of the timing."""

    def __init__(self, ctx, group):
        xl, torch, dist = ctx["xl"], ctx["torch"], ctx["dist"]
        self.ctx = ctx
        rank, world = ctx["rank"], ctx["world"]
        self.name = ("xlating_multi (C host: ncclBroadcast of every super-block on a communication stream) + xlating_batch" if world > 1
                     else "xlating_multi (C host; one GPU: no communicator, the resident super-block is filtered in place) + xlating_batch")
        uid = None
        if world > 1:
            t = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                t.copy_(torch.frombuffer(bytearray(xl.MultiHost.unique_id()), dtype=torch.uint8))
            dist.broadcast(t, src=0)
            uid = bytes(t.cpu().numpy().tobytes())
        self.m = xl.MultiHost(FS, "cu8", BLOCK_BYTES, group_blocks=group, rank=rank, world=world, uid=uid,
                              device=torch.cuda.current_device())
        self.eng = self.m.engine(rank)
        self.k = 0
        self.comm_count = self.m.comm_count()
        if world > 1 and self.comm_count != world:  # (a communicator of another size would be another job's)
            raise RuntimeError(f"RCCL communicator counts {self.comm_count} ranks, the job has {world}")
        if world > 1:
            self.m.feed_timing(True)

    def feed_timing(self):
        """(feeds measured, broadcast ms, hidden ms) since the last read -- HIP events on the communication stream."""
        return self.m.feed_timing_read()

    def add_client(self, c, taps):
        return self.m.add_client(c, D, taps, client_center_freq(c))

    def call(self, mode, nblocks, first_block=0, count=None, advance=True):
        groups = self.ctx["dev_groups"]
        base = groups[self.k % len(groups)].data_ptr() if groups else 0
        last = GROUP if count is None else first_block + count
        for j in range(first_block, last, nblocks):
            self.m.feed(base + j * BLOCK_BYTES if base else 0, BLOCK_BYTES, nblocks, mode)
        if advance:
            self.k += 1

    def sync(self):
        self.m.sync()
        self.ctx["torch"].cuda.synchronize()

    def close(self):
        self.m.close()


def make_host(ctx, group):
    """The C host (include/xlating_multi.h) unless `--feed torch` asks for the torch.distributed feeder.  If any rank cannot
    create the C host (e.g. an RCCL set-up problem) EVERY rank stops with an error: a multi-GPU run that silently measured
    the Python feeder instead would be mistaken for the C host's number."""
    torch, dist, world = ctx["torch"], ctx["dist"], ctx["world"]
    host, err = None, None
    if not ctx["cuda"] or ctx["feed"] == "torch":
        try:
            if os.environ.get("XL_BENCH_FAIL_RANK") == str(ctx["rank"]):  # (tests/test_bench_launch.py: one rank cannot create its engine)
                raise RuntimeError("simulated engine create failure (XL_BENCH_FAIL_RANK)")
            host = TorchHost(ctx, group)
        except Exception as e:  # noqa: BLE001 -- reported below, after every rank has heard of it
            err = e
    elif world > 1 or os.environ.get("XL_BENCH_CHOST_THREAD"):
        # (RCCL set-up is the one step of this path no single-GPU box can exercise: give it a deadline instead of trusting it)
        import threading

        box = {}

        cur = torch.cuda.current_device()

        def create():
            try:
                torch.cuda.set_device(cur)  # (the current device is per thread)
                box["host"] = CHost(ctx, group)
            except Exception as e:  # noqa: BLE001
                box["err"] = e

        th = threading.Thread(target=create, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("XL_BENCH_CHOST_TIMEOUT", "120")))
        torch.cuda.set_device(cur)
        host, err = box.get("host"), box.get("err", "timed out" if th.is_alive() else None)
    else:
        try:
            host = CHost(ctx, group)
        except Exception as e:  # noqa: BLE001 -- reported below
            err = e
    ok = 1 if host is not None else 0
    if world > 1:  # every rank learns whether ALL of them have an engine: one failure stops the whole job, nobody waits in a collective
        t = torch.tensor([ok], dtype=torch.int32, device="cuda" if ctx["cuda"] else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item())
    if ok:
        return host
    if host is not None:
        host.close()
    raise SystemExit(f"bench.py: rank {ctx['rank']}: the engine host could not be created on every rank (here: {err if err else 'ok, another rank failed'}); "
                     "nothing was measured.  (C host failures: `--feed torch` runs the torch.distributed feeder instead and says so in config.feed.)")


def run_workload(ctx, total_clients, ntaps_rate, steps, warmup, mode, group=GROUP, options=None, staggered=False,
                 spot=False, poly3=True, blocks_per_step=BLOCKS_PER_STEP, repeats=1, replay_calls=0):
    """Build this rank's engine with its shard of clients and time `repeats` x `steps` steps.  Returns dict of measurements.
    options: engine plan options (xlating_batch_set_option); staggered: the clients join over 21 consecutive blocks
    before the warm-up (every join lands on another output grid) instead of all before block 0.
    replay_calls > 0: no timing at all -- warm up, run that many calls, return (the PMC passes profile this)."""
    xl, torch, dist, rank, world = ctx["xl"], ctx["torch"], ctx["dist"], ctx["rank"], ctx["world"]
    cuda = ctx["cuda"]
    code, taps = ctx["lpf"](1.0, FS, RATE // 2, RATE // ntaps_rate)
    assert code == 0
    mine = shard_clients(total_clients, world, rank)
    host = make_host(ctx, group)
    eng = host.eng
    if cuda:
        for k, v in (options or {}).items():
            eng.set_option(k, v)
    calls_per_step = blocks_per_step // GROUP
    ids = {}

    def call(nblocks=group):
        host.call(mode, nblocks)

    if staggered and cuda:  # one block per call while the clients trickle in: 21 joins, then 3 blocks to mature + merge
        per = -(-max(len(mine), 1) // 21)
        for blk in range(24):
            for c in mine[blk * per:(blk + 1) * per]:
                ids[c] = host.add_client(c, taps)
            host.call(mode, 1, first_block=blk % GROUP, count=1, advance=(blk % GROUP == GROUP - 1))
    else:
        for c in mine:
            ids[c] = host.add_client(c, taps)

    if replay_calls:
        for _ in range(4 + replay_calls):
            call()
        host.sync()
        host.close()
        return None

    for _ in range(warmup * calls_per_step if warmup else 2):
        call()
    host.sync()
    eng.timing_stride(TIMING_STRIDE)
    eng.timing(True)
    if hasattr(host, "feed_timing") and world > 1:
        host.feed_timing()  # (reset: the warm-up's feeds do not count)
    secs, rank_secs = [], []
    for _ in range(repeats):
        if world > 1:
            dist.barrier()
        host.sync()
        t0 = time.perf_counter()
        for _ in range(steps * calls_per_step):
            call()
        host.sync()
        if world > 1:
            dist.barrier()
        host.sync()
        dt = time.perf_counter() - t0
        if world > 1:
            rank_secs.append(gather_seconds(dist, torch, dt, "cuda" if cuda else "cpu", world))
            dt = reduce_max_seconds(dist, torch, dt, "cuda" if cuda else "cpu")
        secs.append(dt)
    feed = None
    if hasattr(host, "feed_timing") and world > 1:
        nfeed, bms, hms = host.feed_timing()
        if nfeed > 0:
            feed = {"feeds_measured": nfeed, "broadcast_us": round(bms / nfeed * 1e3, 2), "hidden_us": round(hms / nfeed * 1e3, 2),
                    "feed_overlap_frac": round(hms / bms, 4) if bms > 0 else None,
                    "how": "HIP events on the communication stream around every ncclBroadcast (this rank); hidden = the part of a "
                           "broadcast that elapsed while the previous feed's launches were still running on the compute stream"}
    nt, fir_ms, nco_ms = eng.timing_read(reset=True)
    eng.timing(False)
    dt = sorted(secs)[len(secs) // 2]
    plan = eng.describe()
    polyphase = mode == "optimized" and "polyphase: none" not in plan
    klen = eng.output_len(ids[mine[0]]) if mine else 0

    spot_res = None
    if spot and cuda and mine and group in (1, GROUP) and not staggered and world == 1:
        spot_res = parity_spot(ctx, eng, ids, mine, taps, mode, host.k, call, last_blocks=group)

    kernels_ms = None
    if polyphase and poly3 and cuda:  # separate durations of the three launches: a short extra pass, OUTSIDE the timed region
        eng.timing_stride(1)
        eng.timing(2)
        for _ in range(16):
            call()
        host.sync()
        n3, ms3 = eng.timing_polyphase(reset=True)
        eng.timing(False)
        if n3 > 0:
            mix_name = "xlp_mix_mfma_kernel" if "mix=mfma" in plan else "xlp_mix_f32_kernel"
            inv_name = "xlp_inverse8_kernel" if "inv=lanes8" in plan else ("xlp_inverse32_kernel" if "inv=cut32" in plan else "xlp_inverse_kernel")
            kernels_ms = {"xlp_forward_kernel": round(ms3[0] / n3, 4), mix_name: round(ms3[1] / n3, 4), inv_name: round(ms3[2] / n3, 4)}
    feed_name = host.name
    host.close()
    return {"feed": feed_name, "ntaps": int(taps.size), "seconds": dt, "repeat_seconds": secs, "call_ms_avg": fir_ms / max(nt, 1),
            "nco_ms_avg": nco_ms / max(nt, 1), "blocks_per_step": blocks_per_step,
            "timed_calls": nt, "clients_this_rank": len(mine), "total_clients": total_clients, "K_call": int(klen),
            "plan": plan, "polyphase": polyphase, "kernels_ms": kernels_ms, "group": group, "steps": steps,
            "parity_spot": spot_res, "rank_seconds": rank_secs, "feed_timing": feed,
            "rccl_comm_count": getattr(host, "comm_count", None)}


def config5_center_freq(c, n):
    return -4000000 + (8000000 // n) * c


def run_config5(ctx, nclients, steps, spot=True, blocks_per_step=320, replay_calls=0, options=None):
    """BASELINE configs[4] on this GPU: `nclients` clients x 100 kHz off one 10 Msps cf32 stream (D = 100, 257 taps), GROUP blocks of
    131072 samples per call on the engine's own stream, inputs resident in HBM.  Before the timed region EVERY client of one whole
    call is compared with the oracle population (after a first call that loads every filter's history and phase)."""
    import siggen

    xl, torch = ctx["xl"], ctx["torch"]
    taps = siggen.hamming_sinc(C5_TAPS, C5_CUTOFF)
    nel = 2 * S  # float32 elements per block
    x = np.concatenate([(siggen.xs_s16(300 + k, GROUP * nel).astype(np.float32) / np.float32(32768)) for k in range(2)]).astype(np.float32)
    dev = [torch.from_numpy(x[k * GROUP * nel:(k + 1) * GROUP * nel]).cuda() for k in range(2)]
    eng = xl.BatchEngine(C5_FS, "cf32", nel, device=torch.cuda.current_device(), group_blocks=GROUP)
    for k, v in (options or {}).items():
        eng.set_option(k, v)
    fcs = [config5_center_freq(c, nclients) for c in range(nclients)]
    ids = [eng.add_client(C5_D, taps, fc) for fc in fcs]
    ncall = [0]

    def call():
        eng.process_device_group(dev[ncall[0] % 2].data_ptr(), nel, GROUP, "optimized", "engine")
        ncall[0] += 1

    if replay_calls:
        for _ in range(4 + replay_calls):
            call()
        eng.sync()
        eng.close()
        return None
    spot_res = None
    call()
    call()
    if spot:
        odir = os.path.join(ROOT, "oracle")  # the checker (test infrastructure): never on the timed path
        if odir not in sys.path:
            sys.path.insert(0, odir)
        from pyoracle import population

        t0 = time.perf_counter()
        eng.fetch()
        want = population(C5_D, taps, fcs, C5_FS, nel, "cf32", x, GROUP, nwarm=GROUP)
        worst, bad = 0.0, 0
        for c, w in zip(ids, want):
            got = eng.output(c)
            e = float(np.abs(got.astype(np.complex128) - w).max() / np.abs(w).max()) if got.shape == w.shape else float("inf")
            worst, bad = max(worst, e), bad + (0 if e <= 1e-5 else 1)
        spot_res = {"clients": nclients, "clients_failing": bad, "outputs_compared_per_client": int(len(want[0])), "max_rel": worst,
                    "tolerance": 1e-5, "ok": bool(worst <= 1e-5), "seconds": round(time.perf_counter() - t0, 2),
                    "how": "every client of the second call vs oracle/population.c on the host cores (the first call loads history and phase)"}
    calls_per_step = blocks_per_step // GROUP
    for _ in range(calls_per_step):
        call()
    eng.sync()
    eng.timing_stride(TIMING_STRIDE)
    eng.timing(True)
    t0 = time.perf_counter()
    for _ in range(steps * calls_per_step):
        call()
    eng.sync()
    dt = time.perf_counter() - t0
    nt, fir_ms, _ = eng.timing_read(reset=True)
    eng.timing(False)
    plan = eng.describe()
    klen = eng.output_len(ids[0])
    kernels_ms = None
    if "polyphase: none" not in plan:
        eng.timing_stride(1)
        eng.timing(2)
        for _ in range(16):
            call()
        eng.sync()
        n3, ms3 = eng.timing_polyphase(reset=True)
        eng.timing(False)
        if n3 > 0:
            mix_name = "xlp_mix_mfma_kernel" if "mix=mfma" in plan else "xlp_mix_f32_kernel"
            inv_name = "xlp_inverse8_kernel" if "inv=lanes8" in plan else ("xlp_inverse32_kernel" if "inv=cut32" in plan else "xlp_inverse_kernel")
            kernels_ms = {"xlp_forward_kernel": round(ms3[0] / n3, 4), mix_name: round(ms3[1] / n3, 4), inv_name: round(ms3[2] / n3, 4)}
    eng.close()
    blocks = steps * blocks_per_step
    return {"value": nclients * S * blocks / dt / 1e6, "us_per_block": dt / blocks * 1e6, "call_ms_avg": fir_ms / max(nt, 1), "K_call": int(klen),
            "plan": plan, "kernels_ms": kernels_ms, "parity_spot": spot_res, "clients": nclients, "steps": steps, "blocks_per_step": blocks_per_step}


def config5_entry(m5, pmc5):
    """variants[...] entry of BASELINE configs[4] with its own roofline block."""
    n, K = m5["clients"], m5["K_call"]
    units = n * S * GROUP
    call_s = m5["call_ms_avg"] * 1e-3
    fpu = 8.0 * C5_TAPS / C5_D + 6.0 / C5_D
    shared = units * (8.0 / n + 8.0 / C5_D)
    e = {"value": round(m5["value"], 1), "us_per_block": round(m5["us_per_block"], 3), "launches_ms_per_call": round(m5["call_ms_avg"], 4),
         "blocks_per_call": GROUP, "plan": m5["plan"], "parity_spot": m5["parity_spot"], "kernels_ms_per_call": m5["kernels_ms"],
         "workload": f"{n} clients x 100 kHz off one 10 Msps cf32 stream, {2 * S * 4}-byte blocks of {S} complex samples, D={C5_D}, {C5_TAPS} explicit "
                     "Hamming-sinc taps, process_optimized_cf32_cf32 semantics (the cf32-input extension: SURVEY D4; the reference's Airspy "
                     "path feeds the same arithmetic through process_optimized_cs16_cf32, src/xlating.c:374-382)",
         "note": "BASELINE configs[4] ('HBM-roofline run'): 96 kHz is not an integer decimation of 10 Msps (the server rejects it, "
                 "src/tcp_server.c:101-105) -> 100 kHz; 257 taps (computeNtaps never returns an even count)"}
    rl = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel_ms": round(m5["call_ms_avg"], 4), "units_per_launch": units,
          "algorithmic": {"per_client_read_bytes_per_unit": round(8.0 + 8.0 / C5_D, 4), "shared_read_bytes_per_unit": round(8.0 / n + 8.0 / C5_D, 5),
                          "shared_read_bytes_per_call": int(shared), "flop_per_unit_direct_form": round(fpu, 2)}}
    if call_s > 0:
        rl["frac_algorithmic_shared"] = round(shared / call_s / 1e9 / HBM_PEAK_GBS, 4)
        rl["frac_fp32_direct_form"] = round(units * fpu / call_s / 1e12 / FP32_PEAK_TFLOPS, 4)
    if pmc5 and call_s > 0:
        gbs = pmc5["bytes_per_call"] / call_s / 1e9
        rl.update({"achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": pmc5["bytes_per_call"], "traffic_source": "measured in this run",
                   "counter_scope": COUNTER_SCOPE, "traffic_over_shared_read_minimum": round(pmc5["bytes_per_call"] / shared, 2),
                   "frac_is": "HBM bytes of one call's launches by the PMC counters (measured in this run) / the HIP-event duration of those "
                              "launches in this variant's timed region / peak"})
        pk = {}
        mm = re.search(r" V(\d+) M(\d+)", m5["plan"])
        V, M = (int(mm.group(1)), int(mm.group(2))) if mm else (126, 128)
        nseg = -(-(K + 2) // V)
        for kname, ms_ev in (m5["kernels_ms"] or {}).items():
            kpre = kname[:-len("_kernel")]
            pkk = next((v for k, v in pmc5["per_kernel"].items() if k.startswith(kpre) and ("mfma" in k) == ("mfma" in kname)), None)
            ms_k = (pkk or {}).get("ms_per_dispatch_kernel_trace") or ms_ev
            pk[kname] = {"ms": ms_k, "ms_hip_events": ms_ev, "hbm_bytes": (pkk or {}).get("hbm_bytes_per_call"),
                         "frac_hbm": round(pkk["hbm_bytes_per_call"] / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if pkk and ms_k else None}
            if kname.startswith("xlp_mix") and ms_k:
                passes = -(-nseg // 16)
                if "mfma" in kname:
                    # two-half mix: half-precision flops the launch EXECUTES = 3 products x (32 x 32 x 16 x 2) per k-block of 8 branches, per
                    # (bin, 32 columns, pass of 16 segments)
                    pk[kname]["frac_mfma_f16"] = round(3.0 * 32 * 32 * 16 * 2 * -(-C5_D // 8) * M * -(-n // 32) * passes / (ms_k * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)
                else:
                    # float32 mix: one v_mfma_f32_32x32x2_f32 (4096 flop) per (branch, bin, 32 columns, pass)
                    pk[kname]["frac_matrix_f32"] = round(4096.0 * C5_D * M * -(-n // 32) * passes / (ms_k * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)
                    pk[kname]["frac_fp32_useful"] = round(8.0 * n * nseg * M * C5_D / (ms_k * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)
        rl["per_kernel"] = pk
    else:
        rl["traffic"] = None
    e["roofline"] = rl
    return e


def _oracle_population():
    odir = os.path.join(ROOT, "oracle")  # the checker (test infrastructure): never on a timed path
    if odir not in sys.path:
        sys.path.insert(0, odir)
    from pyoracle import population

    return population


def _rel_err(got, want):
    if got.shape != want.shape or want.size == 0:
        return float("inf")
    return float(np.abs(got.astype(np.complex128) - want).max() / np.abs(want).max())


def run_host_delivered(ctx, nclients, ntaps_rate, calls=12, spot=True):
    """What the reference's dsp_worker does with every block (src/dsp_worker.c:57-77: process, then write the output): host blocks
    in (xlating_batch_process_host_group: pinned staging + H2D), every client's outputs back on the host (xlating_batch_fetch: D2H of
    the whole output image) before the next call -- synchronous, PCIe-inclusive, never `value`.  After the timed loop every client's
    delivered outputs of the last call are compared with the oracle population."""
    xl = ctx["xl"]
    code, taps = ctx["lpf"](1.0, FS, RATE // 2, RATE // ntaps_rate)
    eng = xl.BatchEngine(FS, "cu8", BLOCK_BYTES, device=ctx["torch"].cuda.current_device(), group_blocks=GROUP)
    for c in range(nclients):
        eng.add_client(D, taps, client_center_freq(c))
    xs = [make_group(g) for g in range(2)]
    for k in range(2):
        eng.process_host_group(xs[k % 2], GROUP, "optimized")
        eng.fetch()
    t0 = time.perf_counter()
    for k in range(calls):
        eng.process_host_group(xs[k % 2], GROUP, "optimized")
        eng.fetch()
    dt = time.perf_counter() - t0
    out_bytes = sum(eng.output_len(c) for c in range(nclients)) * 8
    spot_res = None
    if spot:  # (calls + 2 calls so far; the last one filtered xs[(calls + 1) % 2] after xs[calls % 2])
        done = calls + 2
        x = np.concatenate([xs[(done - 2) % 2], xs[(done - 1) % 2]])
        want = _oracle_population()(D, taps, [client_center_freq(c) for c in range(nclients)], FS, BLOCK_BYTES, "cu8", x, GROUP, nwarm=GROUP,
                                    skip_fresh=S, skip_calls=(done - 2) * GROUP)
        errs = [_rel_err(eng.output(c), w) for c, w in enumerate(want)]
        spot_res = {"clients": nclients, "clients_failing": int(sum(e > 1e-5 for e in errs)), "max_rel": max(errs), "ok": bool(max(errs) <= 1e-5)}
    eng.close()
    return {"value": round(nclients * S * GROUP * calls / dt / 1e6, 1), "us_per_block": round(dt / (calls * GROUP) * 1e6, 2), "blocks_per_call": GROUP,
            "host_bytes_in_per_call": GROUP * BLOCK_BYTES, "host_bytes_out_per_call": int(out_bytes),
            "pcie_GBs_out": round(out_bytes * calls / dt / 1e9, 2), "parity_spot": spot_res,
            "note": "process_host_group + fetch per call, synchronous: every client's outputs of every block delivered to host memory "
                    "(src/dsp_worker.c:74-77 writes them out).  PCIe-inclusive; the headline keeps outputs in HBM"}


def config3_clients(ctx):
    """BASELINE configs[2] as SURVEY 8(d) defines it: 32 x 48 kHz (D = 42, 505 taps) + 32 x 96 kHz (D = 21, 253 taps), fc = -900 kHz + c x 28 kHz."""
    t48 = ctx["lpf"](1.0, FS, 24000, 9600)[1]
    t96 = ctx["lpf"](1.0, FS, 48000, 19200)[1]
    return [((42, t48) if c % 2 == 0 else (21, t96)) + (-900000 + c * 28000,) for c in range(64)]


def run_config3_mixed(ctx, steps=40, spot=True):
    """BASELINE configs[2]: 64 concurrent clients at mixed 48 / 96 kHz sharing one 2.016 Msps block stream, GROUP blocks per call, inputs
    resident in HBM, engine's own stream.  Every client of the call after the timed loop is compared with the oracle population."""
    xl, torch = ctx["xl"], ctx["torch"]
    cl = config3_clients(ctx)
    eng = xl.BatchEngine(FS, "cu8", BLOCK_BYTES, device=torch.cuda.current_device(), group_blocks=GROUP)
    ids = [eng.add_client(dd, tt, fc) for dd, tt, fc in cl]
    xs = [make_group(g) for g in range(2)]
    dev = [torch.from_numpy(x).cuda() for x in xs]
    n = [0]

    def call():
        eng.process_device_group(dev[n[0] % 2].data_ptr(), BLOCK_BYTES, GROUP, "optimized", "engine")
        n[0] += 1

    for _ in range(4):
        call()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    eng.sync()
    dt = time.perf_counter() - t0
    plan = eng.describe()
    spot_res = None
    if spot:
        done = n[0]
        call()
        call()
        eng.fetch()
        x = np.concatenate([xs[done % 2], xs[(done + 1) % 2]])
        errs = {}
        for dd in (42, 21):
            sel = [i for i, c in enumerate(cl) if c[0] == dd]
            want = _oracle_population()(dd, cl[sel[0]][1], [cl[i][2] for i in sel], FS, BLOCK_BYTES, "cu8", x, GROUP, nwarm=GROUP,
                                        skip_fresh=S, skip_calls=done * GROUP)
            for i, w in zip(sel, want):
                errs[i] = _rel_err(eng.output(ids[i]), w)
        worst = max(errs.values())
        spot_res = {"clients": len(cl), "clients_failing": int(sum(e > 1e-5 for e in errs.values())), "max_rel": worst, "ok": bool(worst <= 1e-5)}
    eng.close()
    blocks = steps * GROUP
    return {"value": round(len(cl) * S * blocks / dt / 1e6, 1), "us_per_block": round(dt / blocks * 1e6, 3), "blocks_per_call": GROUP,
            "plan": plan, "parity_spot": spot_res,
            "workload": "BASELINE configs[2]: 32 x 48 kHz (D=42, 505 taps) + 32 x 96 kHz (D=21, 253 taps) clients off one 2.016 Msps cu8 stream"}


def run_config2_dropin(ctx, calls=300, spot=True):
    """BASELINE configs[1]: ONE client through the drop-in C API (include/xlating.h: create / process_optimized_cu8_cf32 / destroy), host
    block in, host-visible output back per call -- what dsp_worker.c:57-77 does; latency-bound (us per 262144-byte block).  The outputs
    of three further calls are compared with one oracle filter fed the same blocks from the start."""
    import gc

    xl = ctx["xl"]
    taps = ctx["lpf"](1.0, FS, RATE // 2, 9600)[1]
    xs = [make_group(g)[:BLOCK_BYTES] for g in range(2)]
    f = xl.XlatingFilter(D, taps, -12000, FS, BLOCK_BYTES)
    done = 0  # blocks the filter has seen: block k of its stream is xs[k % 2]
    for _ in range(20):
        f.process("optimized", "cu8", "cf32", xs[done % 2])
        done += 1
    gc_was = gc.isenabled()
    gc.disable()  # (a generation-2 collection inside the ctypes wrapper is a 1-35 ms outlier: tools/dropin_python_stall.py)
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        f.process("optimized", "cu8", "cf32", xs[done % 2])
        ts.append(time.perf_counter() - t0)
        done += 1
    if gc_was:
        gc.enable()
    spot_res = None
    if spot:
        odir = os.path.join(ROOT, "oracle")
        if odir not in sys.path:
            sys.path.insert(0, odir)
        from pyoracle import Oracle

        # one oracle filter: stream state fast-forwarded over all blocks but the last, that one filtered for real (it loads the sample
        # history), then three further blocks through both
        o = Oracle(D, taps, -12000, FS, BLOCK_BYTES)
        o.skip_calls(S, done - 1)
        o.process("cu8", xs[(done - 1) % 2])
        worst = 0.0
        for _ in range(3):
            worst = max(worst, _rel_err(f.process("optimized", "cu8", "cf32", xs[done % 2]), o.process("cu8", xs[done % 2])))
            done += 1
        o.close()
        spot_res = {"clients": 1, "clients_failing": int(worst > 1e-5), "max_rel": worst, "ok": bool(worst <= 1e-5)}
    f.close()
    ts.sort()
    mean = sum(ts) / len(ts)
    return {"value": round(S / mean / 1e6, 1), "us_per_block": round(mean * 1e6, 2), "median_us": round(ts[len(ts) // 2] * 1e6, 2),
            "p99_us": round(ts[int(len(ts) * 0.99) - 1] * 1e6, 2), "blocks_per_call": 1, "parity_spot": spot_res,
            "workload": "BASELINE configs[1]: one client, create_frequency_xlating_filter + process_optimized_cu8_cf32 per 262144-byte host block, 505 taps, D=42"}


def parity_spot(ctx, eng, ids, mine, taps, mode, calls_done, call, last_blocks=GROUP):
    """EVERY client of the engine the timed region just ran, vs a population of oracle filters on the host cores
    (oracle/population.c, one reference-model filter per client over pthreads).  The oracles' stream state (phase
    recurrence with per-block renormalisation, history counter) is fast-forwarded over the blocks processed so far
    without filtering; then they filter the next super-block for real to load their sample history, and the one after is
    compared (native: bit-exact; optimized: max|d| / max|y| <= 1e-5 per client)."""
    odir = os.path.join(ROOT, "oracle")  # the checker (test infrastructure): never on the timed path
    if odir not in sys.path:
        sys.path.insert(0, odir)
    from pyoracle import population

    t0 = time.perf_counter()
    x = np.concatenate([make_group((calls_done + rnd) % NSRC_GROUPS) for rnd in range(2)])
    call()
    call()
    eng.fetch()
    # (an engine driven one block per call holds the outputs of the last BLOCK only: last_blocks = 1)
    want = population(D, taps, [client_center_freq(c) for c in mine], FS, BLOCK_BYTES, "cu8", x, last_blocks, nwarm=2 * GROUP - last_blocks,
                      skip_fresh=S, skip_calls=calls_done * GROUP)
    t_oracle = time.perf_counter() - t0
    worst, exact, want_len, bad = 0.0, True, 0, 0
    for c, w in zip(mine, want):
        got = eng.output(ids[c])
        if got.shape != w.shape:
            worst, exact, bad = float("inf"), False, bad + 1
            continue
        want_len = len(w)
        e = float(np.abs(got.astype(np.complex128) - w).max() / np.abs(w).max())
        same = np.array_equal(got.view(np.uint8), w.view(np.uint8))
        worst, exact = max(worst, e), exact and same
        bad += 0 if (same if mode == "native" else e <= 1e-5) else 1
    return {"clients": len(mine), "clients_failing": bad, "blocks_before": calls_done * GROUP, "outputs_compared_per_client": want_len,
            "max_rel": worst, "bit_exact": bool(exact), "tolerance": 0.0 if mode == "native" else 1e-5,
            "ok": bool(exact if mode == "native" else worst <= 1e-5), "seconds": round(time.perf_counter() - t0, 2),
            "oracle_seconds": round(t_oracle, 2),
            "how": "every client of this GPU vs oracle/population.c on the host cores: oracles fast-forwarded over the run's blocks "
                   "(phase recurrence + per-block renormalisation only), one super-block filtered to load the history, the next one compared"}


def cpu_baseline(ntaps_rate, seconds=12.0):
    """Reference (oracle/_ref/libref_fast.so) -- or the repo's CPU restatement when that build is absent -- on the
    host cores, via the native pthread driver oracle/cpu_bench: one thread per client, each with its own filter
    over the same block (the reference's dsp_worker model, src/dsp_worker.c:41-88).  Bounded sample: `seconds` of
    wall time split between a 1-thread and an all-threads run."""
    odir = os.path.join(ROOT, "oracle")
    drv = os.path.join(odir, "cpu_bench")
    flags = open("/proc/cpuinfo").read()
    model = next((l.split(":", 1)[1].strip() for l in flags.splitlines() if l.startswith("model name")), "unknown")
    flat = " " + flags.replace("\n", " ") + " "
    ref = os.path.join(odir, "_ref", "libref_fast.so")
    if os.path.exists(ref) and " avx2 " in flat and " fma " in flat:
        lib, api, variant, kind = ref, "ref", "optimized", "reference"
        what = "unmodified src/xlating.c process_optimized_cu8_cf32 (hand-written AVX path, no phase renormalisation; gcc -O3 -ffast-math -mavx2 -mfma)"
    else:
        lib, api, variant, kind = os.path.join(odir, "liboracle.so"), "orc", "native", "port"
        what = "oracle/xlating_oracle.c scalar restatement (gcc -O2, canonical order)"
    threads = os.cpu_count() or 1
    phys = set()
    pid = cid = None
    for line in flags.splitlines():
        if line.startswith("physical id"):
            pid = line.split(":")[1].strip()
        elif line.startswith("core id"):
            cid = line.split(":")[1].strip()
        elif not line.strip() and pid is not None and cid is not None:
            phys.add((pid, cid))
            pid = cid = None
    cores = len(phys) or threads

    def run(nthreads, secs):
        r = subprocess.run([drv, lib, api, variant, str(nthreads), str(secs), str(FS), str(RATE), str(RATE // ntaps_rate),
                            str(BLOCK_BYTES)], capture_output=True, text=True, timeout=secs * 3 + 60)
        if r.returncode != 0:
            raise RuntimeError(f"cpu_bench failed: {r.stderr[-500:]}")
        return json.loads(r.stdout.strip().splitlines()[-1])

    one = run(1, max(2.0, seconds * 0.3))
    allc = run(threads, max(3.0, seconds * 0.7))
    return {
        "value": round(allc["msps"], 1), "unit": "Msamples/s", "cores": cores, "threads": threads, "kind": kind,
        "single_thread_value": round(one["msps"], 1),
        "sample_short": f"{'reference AVX2 -O3 -ffast-math build' if kind == 'reference' else 'scalar oracle port'}, {allc['ntaps']} taps D={D}, one filter per thread over a shared "
                        f"{BLOCK_BYTES}-byte block: {allc['calls']} calls / {threads} threads / {allc['seconds']:.1f} s (+ {one['seconds']:.1f} s on 1 thread); {model}",
        "sample": f"{what}; lpf_cutoff_rate={ntaps_rate} -> {allc['ntaps']} taps, D={D}, {BLOCK_BYTES}-byte cu8 blocks, one "
                  f"filter per thread over a shared block; {allc['calls']} calls on {threads} threads ({cores} physical cores) in "
                  f"{allc['seconds']:.1f} s wall (+ {one['calls']} calls on 1 thread in {one['seconds']:.1f} s); host CPU: {model}",
    }


def polyphase_traffic_model(nclients, K_call, ntaps, group, M=128, mfma=True):
    """HBM bytes one CALL of `group` blocks moves by design on the polyphase path (xl_polyphase.h), per GPU: branch
    spectra R read once per call (8 D M bytes per client), mixed spectra Y written and read back (8 M bytes per client
    and segment), outputs written, NCO phase table (every 16th phase) written and read; the shared spectra X and the
    raw blocks are noise."""
    A = -(-ntaps // D)
    V = M - A + 1
    nseg = -(-(K_call + 2) // V)
    # (operand-form images: 8 bytes per (branch, bin) -- two halves per component, or two float32 factors --, 8 branches per k-block)
    dpad = -(-D // 8) * 8
    per_client = 8 * dpad * M + 2 * 8 * M * nseg + 8 * K_call + 2 * 8 * (K_call // 16)
    return {"transform_length_M": M, "blocks_per_call": group, "bytes_per_call": int(nclients * per_client),
            "bytes_per_block": int(nclients * per_client / group), "bytes_per_client_per_block": int(per_client / group),
            "R_branch_spectra_per_call": 8 * dpad * M, "Y_mixed_spectra_write_plus_read_per_call": 2 * 8 * M * nseg,
            "out_per_call": 8 * K_call, "phase_table_write_plus_read_per_call": 2 * 8 * (K_call // 16)}


def summarize(m):
    """value and model figures from the timed region of `m` (HIP-event duration of the calls' launches in it)."""
    blocks = m["steps"] * m["blocks_per_step"]
    value = m["total_clients"] * S * blocks / m["seconds"] / 1e6
    units_per_call = m["clients_this_rank"] * S * m["group"]
    bpu = algorithmic_bytes_per_unit(D)
    fpu = flops_per_unit(m["ntaps"], D)
    call_s = m["call_ms_avg"] * 1e-3
    ach_gbs = units_per_call * bpu / call_s / 1e9 if call_s > 0 else 0.0
    ach_tf = units_per_call * fpu / call_s / 1e12 if call_s > 0 else 0.0
    return value, ach_gbs, ach_tf, bpu, fpu


def variant_entry(m, note=None):
    v, g, t, _, f = summarize(m)
    e = {"value": round(v, 1), "ms_per_step": round(m["seconds"] / m["steps"] * 1e3, 4),
         "us_per_block": round(m["seconds"] / (m["steps"] * m["blocks_per_step"]) * 1e6, 3),
         "launches_ms_per_call": round(m["call_ms_avg"], 4), "blocks_per_call": m["group"],
         "path": "polyphase" if m["polyphase"] else "direct FIR kernel",
         "model_hbm_frac": round(g / HBM_PEAK_GBS, 4),
         "fp32_frac": None if m["polyphase"] else round(t / FP32_PEAK_TFLOPS, 4),
         "achieved_TFLOPs": None if m["polyphase"] else round(t, 2), "plan": m["plan"]}
    if m.get("rank_seconds"):
        e["per_rank_seconds"] = m["rank_seconds"]
        e["feed_timing"] = m.get("feed_timing")
        e["rccl_comm_count"] = m.get("rccl_comm_count")
    if note:
        e["note"] = note
    return e


# what FETCH_SIZE / WRITE_SIZE count (MI355X_MICROARCH.md, HBM section; profiles/r04_mall_calibration.txt)
COUNTER_SCOPE = ("L2 <-> fabric requests (TCC_EA read / write requests x request size): Infinity-Cache (MALL) hits are included, so "
                 "`traffic` is an UPPER bound of the HBM bytes and `frac` an upper bound of the HBM utilisation")

PMC_KERNEL_PREFIXES = ("xlp_forward", "xlp_mix", "xlp_inverse", "xlp_fused", "xl_fir_kernel", "xl_nco_chain", "xl_nco_table", "xl_update_history")


def _pmc_rows(path, value_of):
    """rows of a rocprofv3 csv in dispatch order: (kernel name without template arguments, value_of(row))"""
    import csv

    rows = []
    for i, row in enumerate(csv.DictReader(open(path))):
        k = row["Kernel_Name"].split("(")[0]
        k = k[5:] if k.startswith("void ") else k
        k = k.split("<")[0]
        if not k.startswith(PMC_KERNEL_PREFIXES):
            continue
        order = int(row.get("Dispatch_Id") or row.get("Start_Timestamp") or i)
        rows.append((order, k, value_of(row)))
    rows.sort(key=lambda r: r[0])
    return rows


def _split_workloads(rows, n):
    """The replay process runs its workloads one after the other, each on a fresh engine.  Only a fresh engine tabulates phases with
    stand-alone xl_nco_table_kernel launches: for its first call, and for its second one (the clients have matured behind the first
    call and the plan is rebuilt) -- every later call finds its table tabulated ahead.  So a table launch marks where a workload
    begins; the few rows between an engine's two table launches (its first call) belong to the same workload."""
    segs = []
    for _, k, v in rows:
        if k.startswith("xl_nco_table") and (not segs or sum(len(x) for x in segs[-1].values()) > 16):
            segs.append({})
        if segs:
            segs[-1].setdefault(k, []).append(v)
    return segs if len(segs) == n else None


def measure_traffic(args, workloads):
    """HBM bytes per call of `workloads` (a list of "server:<clients>" / "config5:<clients>") on THIS box: two rocprofv3 --pmc passes
    (FETCH_SIZE, WRITE_SIZE: the guide's HBM section asks for separate passes) over `bench.py --replay-calls N --replay-set ...`
    (ONE process per pass replays all the workloads back to back), run as subprocesses after the timed region, and a third pass
    without counters for the kernels' own durations.
    gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of wide coalesced reads ->
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB.  Returns ({workload: result dict} | None, note)."""
    import glob
    import shutil
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="xl_bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    tail = [sys.executable, os.path.abspath(__file__), "--replay-calls", str(PMC_REPLAY_CALLS), "--replay-set", ",".join(workloads),
            "--mode", args.mode, "--lpf-cutoff-rate", str(args.lpf_cutoff_rate)]
    tmo = float(os.environ.get("XL_BENCH_PMC_TIMEOUT", "240"))
    segs = {}  # counter -> per workload {kernel: [per dispatch]}
    t0 = time.perf_counter()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            r = subprocess.run([rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + tail,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=tmo)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"
            rows = _pmc_rows(files[0], lambda row, c=counter: float(row["Counter_Value"]) if row["Counter_Name"] == c else None)
            segs[counter] = _split_workloads([x for x in rows if x[2] is not None], len(workloads))
            if segs[counter] is None:
                return None, f"the {counter} pass does not show {len(workloads)} workloads"
        # third pass, no counters: the kernels' own durations as the profiler sees them in the production arrangement (the chain
        # kernel running beside the three launches on its side stream) -- what `rocprofv3 --kernel-trace --stats` prints
        durs = None
        d = os.path.join(out, "trace")
        r = subprocess.run([rocprof, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + tail,
                           cwd="/tmp", env=env, capture_output=True, text=True, timeout=tmo)
        files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode == 0 and files:
            durs = _split_workloads(_pmc_rows(files[0], lambda row: (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6), len(workloads))
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 --pmc pass timed out"
    finally:
        shutil.rmtree(out, ignore_errors=True)
    results = {}
    for wi, wl in enumerate(workloads):
        fs, ws = segs["FETCH_SIZE"][wi], segs["WRITE_SIZE"][wi]
        main_k = next((k for k in fs if k.startswith("xlp_inverse")), None) or next((k for k in fs if k.startswith("xl_fir_kernel")), None)
        if main_k is None:
            return None, f"no engine kernel in the counter output of {wl}"
        per, total = {}, 0.0
        ncalls = max(len(fs[main_k]), 1)
        for k in sorted(set(fs) | set(ws)):
            f, w = fs.get(k, []), ws.get(k, [])
            n = max(len(f), len(w), 1)
            if n <= 2 or k.startswith("xl_nco_table"):  # (a fresh engine's first call -- direct launch, stand-alone tabulation --: not part of a steady call)
                continue
            drop = 3 if n > 6 else 0  # (the first calls: cold caches)
            fm = sum(f[drop:]) / max(len(f[drop:]), 1)
            wm = sum(w[drop:]) / max(len(w[drop:]), 1)
            per_dispatch = (2.0 * fm + wm) * 1024.0
            per_call = per_dispatch * n / ncalls  # (a chain launch covers several calls; every other kernel is once per call)
            per[k] = {"hbm_bytes_per_dispatch": int(per_dispatch), "dispatches_per_call": round(n / ncalls, 3), "hbm_bytes_per_call": int(per_call),
                      "FETCH_SIZE_KiB": round(fm, 1), "WRITE_SIZE_KiB": round(wm, 1)}
            dk = (durs[wi] if durs else {}).get(k, [])
            dk = dk[3:] if len(dk) > 6 else dk
            if dk:
                per[k]["ms_per_dispatch_kernel_trace"] = round(sum(dk) / len(dk), 4)
            total += per_call
        results[wl] = {"bytes_per_call": int(total), "per_kernel": per, "calls_profiled": ncalls, "seconds": round(time.perf_counter() - t0, 1),
                       "correction": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): "
                                     "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
                       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --replay-calls %d --replay-set %s (two passes; "
                                  "one process per pass replays all the listed workloads) + one rocprofv3 --kernel-trace pass without counters for the "
                                  "kernels' durations" % (PMC_REPLAY_CALLS, ",".join(workloads))}
    return results, "measured in this run"


# DESIGN.md section 8 (history: DESIGN_HISTORY.md section 7): the curve this workload is EXPECTED to follow (per-GPU step times measured on one GPU; the 2 MB
# broadcast per call hides behind a 200+ us call).  Strong scaling saturates at once: below ~1000 clients per GPU every GPU
# is bound by the float32 phase recurrence (3121 sequential steps per block and client).
EXPECTED_STRONG = {"clients_total": 1024, "Msamples_per_s": {"1": 5.39e6, "2": 5.64e6, "4": 5.79e6, "8": 5.94e6},
                   "speedup": {"1": 1.0, "2": 1.05, "4": 1.07, "8": 1.10},
                   "why": "DESIGN.md section 8 (history: DESIGN_HISTORY.md section 7): one GPU already runs 1024 clients at 24.9 us per block, within 10 % of the floor the NCO "
                          "phase recurrence sets for ANY client count (3121 dependent float32 steps per block and client: 22.6 us); 1024 "
                          "clients IN TOTAL leave 512 / 256 / 128 per GPU (23.8 / 23.2 / 22.6 us per block measured on one GPU), so "
                          "adding GPUs only pays with MORE clients (weak scaling)"}
EXPECTED_WEAK = {"clients_per_gpu": 1024, "Msamples_per_s_per_gpu": "5.3e6 - 5.5e6", "efficiency": "~1.0 (no data-path collective besides one 2 MB broadcast per call)"}


COMPACT_MAX_BYTES = 4096  # the driver keeps an 8 KB tail of stdout: the one line it parses must fit with room to spare
FULL_JSON_DEFAULT = os.path.join(ROOT, "profiles", "bench_last_full.json")

# compact `configs` keys <- prefix of the full record's variants[...] key
CONFIG_KEYS = (("one_block_per_call", "one block per call"), ("config3_64_mixed_clients", "config 3"), ("config2_dropin_single_filter", "config 2"),
               ("clients_2048", "2048 clients"), ("clients_4096", "4096 clients"), ("config5_cf32_10msps_1024_clients", "config 5"),
               ("all_f32", "polyphase, float32 matrix-core mix"), ("host_delivered", "host-delivered outputs"))


def _sig(x, n=4):
    return None if x is None else (float(f"{x:.{n}g}") if isinstance(x, float) else x)


def _spot_compact(sp):
    if not sp:
        return None
    return {"clients": sp.get("clients"), "clients_failing": sp.get("clients_failing"), "max_rel": _sig(sp.get("max_rel"), 3), "ok": sp.get("ok")}


def compact_line(full):
    """The ONE stdout line: the bench contract's keys + roofline + cpu_baseline + parity + one short entry per BASELINE config, nothing else;
    <= COMPACT_MAX_BYTES.  Everything descriptive (per-kernel tables, plans, notes, expected scaling) stays in the full record
    (`--full-json`, default profiles/bench_last_full.json; also printed to stderr)."""
    rl = full.get("roofline") or {}
    cfg = full.get("config") or {}
    c = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                  "data")}
    c["dtype"] = "f32"
    c["config"] = {"workload": f"{cfg.get('clients_total')} clients x 48 kHz off one 2.016 Msps cu8 stream, D={D}, {cfg.get('ntaps')} taps, {BLOCK_BYTES}-byte blocks "
                               f"(BASELINE configs[3]; N=1: the >=1000-client single-GPU target)",
                   "clients_total": cfg.get("clients_total"), "clients_per_gpu": cfg.get("clients_per_gpu"), "blocks_per_call": cfg.get("blocks_per_call"),
                   "us_per_block": cfg.get("us_per_block"), "mode": cfg.get("mode"), "mix_products": cfg.get("mix_products"),
                   "parallelism": "single GPU" if full.get("n_gpus") == 1 else f"clients c%{full.get('n_gpus')}, one RCCL broadcast per {GROUP} blocks",
                   "rccl_ranks": cfg.get("rccl_ranks"),
                   "feed": ("xlating_multi (C host)" if (cfg.get("feed") or "").startswith("xlating_multi") else
                            "torch.distributed feeder" if (cfg.get("feed") or "").startswith("torch") else "other")}
    r = {k: rl.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "frac_algorithmic_shared", "traffic", "kernel_ms", "units_per_launch",
                                "call_period_ms", "step_bound", "chain_ms_per_call", "frac_of_recurrence_floor")}
    r["kernel"] = rl.get("kernel_short")
    r["traffic_measured_in_run"] = bool(rl.get("traffic_source") == "measured in this run")
    pk = {}
    for k, v in (rl.get("per_kernel") or {}).items():
        pk[k.replace("_kernel", "")] = {"ms": _sig(v.get("ms")), "frac_hbm": v.get("frac_hbm")}
    if pk:
        r["per_kernel"] = pk
    c["roofline"] = r
    cb = full.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "threads", "kind", "single_thread_value")}
        c["cpu_baseline"]["sample"] = cb.get("sample_short")
    c["parity_spot"] = _spot_compact(full.get("parity_spot"))
    configs = {}
    vs = full.get("variants") or {}
    for key, prefix in CONFIG_KEYS:
        v = next((vv for kk, vv in vs.items() if kk.startswith(prefix)), None)
        if isinstance(v, dict) and v.get("value") is not None:
            sp = v.get("parity_spot")
            e = {"value": _sig(float(v["value"]), 5), "us_per_block": _sig(float(v["us_per_block"])), "parity_ok": (sp or {}).get("ok")}
            vr = v.get("roofline") or {}
            if vr.get("frac") is not None:
                e["frac"] = vr["frac"]
            if vr.get("frac_algorithmic_shared") is not None:
                e["frac_shared"] = vr["frac_algorithmic_shared"]
            configs[key] = e
    nat = full.get("native")
    if nat:
        configs["native"] = {"value": _sig(float(nat["value"]), 5), "us_per_block": _sig(float(nat["us_per_block"])),
                             "parity_ok": (nat.get("parity_spot") or {}).get("ok")}
    for kk, vv in vs.items():  # N > 1: the other way to use the GPUs
        if kk.startswith(("strong scaling", "weak scaling")):
            configs[kk.split(" (")[0].replace(" ", "_")] = {"value": _sig(float(vv["value"]), 5), "us_per_block": _sig(float(vv["us_per_block"])), "parity_ok": None}
    if configs:
        c["configs"] = configs
    mg = (full.get("multi_gpu") or {}).get(full.get("scaling")) if full.get("multi_gpu") else None
    if mg:
        ft = mg.get("feed_timing") or {}
        c["multi_gpu"] = {"per_rank_seconds_last_repeat": (mg.get("per_rank_seconds") or [None])[-1], "rccl_comm_count": mg.get("rccl_comm_count"),
                          "broadcast_us": ft.get("broadcast_us"), "feed_overlap_frac": ft.get("feed_overlap_frac")}
    c["full_record"] = full.get("full_record")
    line = json.dumps(c, separators=(",", ":"))
    if len(line) > COMPACT_MAX_BYTES:  # (cannot happen with the fields above; never let a long line cost the driver its record again)
        for k in ("multi_gpu", "configs"):
            c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) <= COMPACT_MAX_BYTES:
                break
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clients", type=int, default=1024, help="weak scaling (default): clients per GPU; strong: in total")
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"])
    ap.add_argument("--mode", default="optimized", choices=["native", "optimized"])
    ap.add_argument("--lpf-cutoff-rate", type=int, default=5, help="server config lpf_cutoff_rate: 5 -> 505 taps, 1 -> 101")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spot", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes (traffic falls back to profiles/pmc_latest.json)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--feed", default="c", choices=["c", "torch"], help="c: include/xlating_multi.h (C host, RCCL); torch: torch.distributed feeder")
    ap.add_argument("--replay-calls", type=int, default=0, help=argparse.SUPPRESS)  # the counter passes profile this: no timing, N calls
    ap.add_argument("--replay-set", default="", help=argparse.SUPPRESS)  # ... of each of these workloads, back to back ("server:1024,config5:1024")
    ap.add_argument("--plumbing-test", action="store_true", help=argparse.SUPPRESS)  # tests/test_bench_launch.py only
    ap.add_argument("--full-json", default=None, help="where the full record goes (default profiles/bench_last_full.json; the stdout line is the compact one)")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, sys.argv[1:]))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (before the HIP runtime comes up: RCCL's IPC needs dmabuf here)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the torch.distributed environment has WORLD_SIZE={world}")
    cuda = not args.plumbing_test
    if cuda and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device (there is no CPU path to time)")
    if cuda:
        torch.cuda.set_device(local_rank)
    dist = None
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if cuda:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        rccl_ranks = dist.get_world_size()

    if cuda:
        import sdr_server_amd as xl

        lpf = xl.create_low_pass_filter
    else:
        xl, lpf = None, (lambda gain, fs, cutoff, tw: (0, np.ones(505 if tw < 20000 else 101, np.float32)))

    dev_groups = []
    if rank == 0 or world == 1:
        dev_groups = [torch.from_numpy(make_group(g)) for g in range(NSRC_GROUPS)]
        if cuda:
            dev_groups = [t.cuda() for t in dev_groups]
    ctx = {"xl": xl, "torch": torch, "dist": dist, "rank": rank, "world": world, "cuda": cuda, "lpf": lpf, "dev_groups": dev_groups,
           "feed": args.feed}

    total_clients = args.clients if args.scaling == "strong" else args.clients * world
    if args.replay_calls:  # what the counter passes profile: these workloads back to back, no timing
        for wl in (args.replay_set.split(",") if args.replay_set else [f"server:{total_clients}"]):
            kind, n = wl.split(":")
            if kind == "config5":
                run_config5(ctx, int(n), 0, spot=False, replay_calls=args.replay_calls)
            else:
                run_workload(ctx, int(n), args.lpf_cutoff_rate, 0, 0, args.mode, replay_calls=args.replay_calls)
        return
    bps = BLOCKS_PER_STEP if cuda else 64
    m = run_workload(ctx, total_clients, args.lpf_cutoff_rate, args.steps, args.warmup, args.mode,
                     spot=not args.no_spot, blocks_per_step=bps, repeats=REPEATS)
    value, model_gbs, ach_tf, bpu, fpu = summarize(m)

    variants = {}
    native = None
    VB = 320  # blocks per step of the variants (their number is context, not the headline)
    if not args.no_variants and (cuda or world > 1):
        vs = max(2, args.steps // 4)
        other_mode = "native" if args.mode == "optimized" else "optimized"
        if cuda:
            mn = run_workload(ctx, total_clients, args.lpf_cutoff_rate, vs, 1, other_mode, spot=not args.no_spot, poly3=False, blocks_per_step=VB)
            native = variant_entry(mn, "the reference's default arithmetic (cpu_optimization NATIVE_CF32, src/config.c:252-264): bit-exact "
                                       "scalar tap order, 4 packed unfused ops per complex MAC, direct FIR kernel")
            native["parity_spot"] = mn["parity_spot"]
        if world > 1:  # the other way to use N GPUs, next to the strong-scaling headline (or vice versa)
            other = "weak" if args.scaling == "strong" else "strong"
            mw = run_workload(ctx, args.clients * world if other == "weak" else args.clients, args.lpf_cutoff_rate, vs, 1, args.mode,
                              poly3=False, blocks_per_step=VB if cuda else 64)
            variants[f"{other} scaling ({args.clients} clients {'per GPU' if other == 'weak' else 'in total'})"] = variant_entry(mw)
        else:
            other_rate = 1 if args.lpf_cutoff_rate != 1 else 5
            mv = run_workload(ctx, total_clients, other_rate, vs, 1, args.mode, poly3=False, blocks_per_step=VB)
            variants[f"lpf_cutoff_rate={other_rate} ({mv['ntaps']} taps)"] = variant_entry(mv)
            m1 = run_workload(ctx, total_clients, args.lpf_cutoff_rate, vs, 1, args.mode, group=1, poly3=False, blocks_per_step=VB, spot=not args.no_spot)
            variants["one block per call (the reference's call granularity)"] = variant_entry(m1)
            variants["one block per call (the reference's call granularity)"]["parity_spot"] = m1["parity_spot"]
            ms = run_workload(ctx, total_clients, args.lpf_cutoff_rate, vs, 1, args.mode, staggered=True, poly3=False, blocks_per_step=VB)
            variants["staggered joins (clients joined over 21 consecutive blocks: 21 output grids, one polyphase class)"] = variant_entry(ms)
            if total_clients >= 8 * 64:  # what each GPU of an 8-GPU strong-scaling run holds (c mod 8)
                m8 = run_workload(ctx, total_clients // 8, args.lpf_cutoff_rate, vs, 1, args.mode, poly3=False, blocks_per_step=VB)
                variants[f"one GPU's share at 8 GPUs ({total_clients // 8} clients): aggregate = 8 x this value minus the feed"] = variant_entry(m8)
            for big in (2048, 4096):  # where the launches, not the NCO recurrence, bound the engine
                if total_clients == 1024:
                    # (EVERY client against the oracle population after the timed region, like the headline)
                    mb = run_workload(ctx, big, args.lpf_cutoff_rate, 2, 1, args.mode, blocks_per_step=VB, spot=not args.no_spot)
                    e = variant_entry(mb)
                    e["kernels_ms_per_call"] = mb["kernels_ms"]
                    e["parity_spot"] = mb["parity_spot"]
                    variants[f"{big} clients on this GPU (kernel-bound regime)"] = e
            if m["polyphase"] and "mix=mfma" in m["plan"]:
                # the same path with every product formed in float32 (option mix_kernel = 3: float32 operands on
                # v_mfma_f32_32x32x2_f32 -- the float32 FMA chain of xlating.c:66-71 -- instead of two-half operands): the all-float32
                # number of this workload, every client against the oracle
                mf = run_workload(ctx, total_clients, args.lpf_cutoff_rate, vs, 1, args.mode, options={"mix_kernel": 3},
                                  spot=not args.no_spot, blocks_per_step=VB)
                e = variant_entry(mf, "float32 operands, float32 FMAs (matrix cores, v_mfma_f32_32x32x2_f32), float32 accumulation in all three launches")
                e["kernels_ms_per_call"] = mf["kernels_ms"]
                e["parity_spot"] = mf["parity_spot"]
                variants["polyphase, float32 matrix-core mix (all-float32 products)"] = e
            if m["polyphase"] and total_clients == 1024:
                # the inverse launch's two kernels in THIS process on THIS box, alternating (box-to-box differences
                # are larger than the difference between them): option inverse_kernel = 6 (the 32 x 4 cut: what the default, 0, picks for
                # launches of this size, xl_polyphase.h: xlp_inverse_pick) / 3 (LDS transform on swizzled rows: round 4's pick there)
                ab = {}
                for big in (2048, 4096):
                    rows = []
                    for rnd in range(2):
                        for inv in (6, 3):
                            mi = run_workload(ctx, big, args.lpf_cutoff_rate, 2, 1, args.mode, options={"inverse_kernel": inv}, blocks_per_step=VB)
                            inv_ms = next((v for k, v in (mi["kernels_ms"] or {}).items() if k.startswith("xlp_inverse")), None)
                            rows.append({"inverse_kernel": inv, "us_per_block": round(mi["seconds"] / (mi["steps"] * mi["blocks_per_step"]) * 1e6, 3),
                                         "inverse_launch_ms_per_call": inv_ms, "launches_ms_per_call": round(mi["call_ms_avg"], 4)})
                    best = {inv: min(r["us_per_block"] for r in rows if r["inverse_kernel"] == inv) for inv in (6, 3)}
                    ab[f"{big} clients"] = {"runs": rows, "best_us_per_block": best, "faster": 6 if best[6] <= best[3] else 3}
                variants["inverse launch A/B in this process (inverse_kernel 6 = the 32 x 4 cut, the size rule's pick here; 3 = LDS transform)"] = ab
                # BASELINE configs[4]: cf32 input at 10 Msps, D = 100, 257 taps (the 'HBM-roofline run'): 1024 clients, every client checked
                m5 = run_config5(ctx, 1024, vs, spot=not args.no_spot, blocks_per_step=VB)
                variants["config 5: cf32 10 Msps, D=100, 257 taps, 1024 clients"] = m5  # (finished below, once the counters are in)
                # what the reference's dsp_worker does with every block: outputs delivered to host memory
                variants["host-delivered outputs (process_host + fetch per call)"] = run_host_delivered(ctx, total_clients, args.lpf_cutoff_rate, spot=not args.no_spot)
                # BASELINE configs[2] (64 mixed clients) and configs[1] (one client through the drop-in C API), re-measured in every run
                variants["config 3: 64 clients at mixed 48 / 96 kHz"] = run_config3_mixed(ctx, spot=not args.no_spot)
                variants["config 2: one client, drop-in process_optimized_cu8_cf32"] = run_config2_dropin(ctx, spot=not args.no_spot)
            if m["polyphase"]:  # the same workload through the direct FIR kernels: the FP32-bound design
                md = run_workload(ctx, total_clients, args.lpf_cutoff_rate, vs, 1, args.mode, options={"polyphase": 0}, poly3=False, blocks_per_step=VB)
                variants[f"process_{args.mode}_cu8_cf32 through the direct FIR kernel ({md['ntaps']} taps)"] = variant_entry(
                    md, "FP32-bound: 96 flop per (client, sample) caps the per-client-read model fraction at "
                        f"{min(1.0, FP32_PEAK_TFLOPS * 1e12 / fpu * bpu / (HBM_PEAK_GBS * 1e9)):.3f}")

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- HBM traffic of one call: counters measured now, on this box; else the committed digest, and the line says so.  One rocprofv3
    # process per counter replays the headline workload, the 2048-client variant (the regime where the launches, not the NCO recurrence,
    # bound the call: north_star's single-GPU target reads ">= 1000 clients at >= 50 % of the rocprof-reported HBM rate") and config 5
    pmc, traffic, traffic_source, per_kernel_bytes = None, None, None, {}
    k5 = next((k for k in variants if k.startswith("config 5")), None)
    kb = next((k for k in variants if k.startswith("2048 clients")), None)
    pmc_all = None
    if cuda and world == 1 and not args.no_pmc:
        wls = [f"server:{total_clients}"] + ([f"server:2048"] if kb and args.clients == 1024 else []) + (["config5:1024"] if k5 else [])
        pmc_all, note = measure_traffic(args, wls)
        if pmc_all:
            pmc = pmc_all[wls[0]]
            traffic, traffic_source = pmc["bytes_per_call"], note
            per_kernel_bytes = {k: v["hbm_bytes_per_call"] for k, v in pmc["per_kernel"].items()}
        else:
            traffic_source = f"in-run counter passes failed ({note}); "
    if kb is not None and variants[kb].get("launches_ms_per_call") and args.clients == 1024 and world == 1 and cuda and not args.no_pmc:
        pmc2 = (pmc_all or {}).get("server:2048")
        if pmc2:
            ms2 = variants[kb]["launches_ms_per_call"]
            gbs2 = pmc2["bytes_per_call"] / (ms2 * 1e-3) / 1e9
            shared2 = 2048 * S * GROUP * (2.0 / 2048 + 8.0 / D)
            variants[kb]["roofline"] = {"bound": "hbm", "achieved": round(gbs2, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(gbs2 / HBM_PEAK_GBS, 4), "traffic": pmc2["bytes_per_call"],
                                        "frac_algorithmic_shared": round(shared2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        "traffic_source": "measured in this run", "kernel_ms": ms2,
                                        "frac_is": "HBM bytes of one call's launches by the PMC counters (measured in this run, 2048 clients) / "
                                                   "the HIP-event duration of those launches in this variant's timed region / peak",
                                        "per_kernel_bytes": {k: v["hbm_bytes_per_call"] for k, v in pmc2["per_kernel"].items()}}
        else:
            variants[kb]["roofline"] = {"traffic": None, "traffic_source": "counter passes failed"}
    if k5 is not None:
        variants[k5] = config5_entry(variants[k5], (pmc_all or {}).get("config5:1024"))
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if traffic is None and os.path.exists(pmc_path) and world == 1 and args.clients == 1024:
        try:
            pj = json.load(open(pmc_path))
            traffic = pj.get("hbm_bytes_per_call_polyphase" if m["polyphase"] else "hbm_bytes_per_call_direct")
            per_kernel_bytes = dict(pj.get("hbm_bytes_kernels_polyphase" if m["polyphase"] else "hbm_bytes_kernels_direct", {}))
            traffic_source = (traffic_source or "") + "replayed from profiles/pmc_latest.json, NOT measured in this run: " + pj.get("source", "?")
        except Exception:
            traffic = None

    call_s = m["call_ms_avg"] * 1e-3
    nloc = m["clients_this_rank"]
    calls_per_step = m["blocks_per_step"] // GROUP
    period_ms = m["seconds"] / args.steps / calls_per_step * 1e3
    tm = None
    if m["polyphase"]:
        import re
        mm = re.search(r"polyphase: cls0 .*? M(\d+)", m["plan"])
        tm = polyphase_traffic_model(nloc, m["K_call"], m["ntaps"], m["group"], int(mm.group(1)) if mm else 256, mfma="mix=mfma" in m["plan"])
    phys_bytes = traffic if traffic else (tm["bytes_per_call"] if tm else None)
    ach = phys_bytes / call_s / 1e9 if phys_bytes and call_s > 0 else 0.0
    roofline = {
        "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(ach / HBM_PEAK_GBS, 4),
        # the USEFUL fraction: SURVEY 8(d)'s shared-read algorithmic bytes (2/N B in + 8/D B out per (client, sample): the block read
        # once per GPU, every output written once) over the same launch time -- what `frac` would be if the path moved nothing else
        "frac_algorithmic_shared": round(nloc * S * m["group"] * (2.0 / max(nloc, 1) + 8.0 / D) / call_s / 1e9 / HBM_PEAK_GBS, 4) if call_s > 0 else None,
        "traffic": traffic, "traffic_source": traffic_source,
        "counter_scope": COUNTER_SCOPE,
        "frac_is": ("HBM bytes of one call's launches by the PMC counters / the HIP-event duration of those launches / peak" if traffic else
                    "NO counter value available: bytes the path moves by design (design_traffic) / launch duration / peak"),
        "kernel_ms": round(m["call_ms_avg"], 4),
        "kernel_ms_note": f"mean HIP-event duration of one call's launches ({m['group']} blocks per call), every {TIMING_STRIDE}th call of the "
                          "timed region, on the launch stream.  A bracketed call also pays for its two event records (~3 us each of stream "
                          "time) and includes the wait for the side-stream phase table when that is late, so this can exceed call_period_ms "
                          "(= timed seconds / calls: what an un-bracketed call occupies) by 1-3 %",
        "call_period_ms": round(period_ms, 4),
        "units_per_launch": nloc * S * m["group"],
        "algorithmic": {"bytes_per_unit": round(bpu, 4), "model": "per-client-read (SURVEY 8(d)): 2 B in + 8/D B out per (client, input sample)",
                        "model_GBs": round(model_gbs, 1), "model_frac": round(model_gbs / HBM_PEAK_GBS, 4),
                        "note": "a MODEL, not traffic: it counts one read of the block per client while the engine reads the block once per GPU "
                                "(shared through L2), so it can exceed 1; shared-read minimum below",
                        "shared_read_bytes_per_unit": round(2.0 / max(nloc, 1) + 8.0 / D, 5),
                        "shared_read_bytes_per_call": int(nloc * S * m["group"] * (2.0 / max(nloc, 1) + 8.0 / D)),
                        "traffic_over_shared_read_minimum": round(traffic / (nloc * S * m["group"] * (2.0 / max(nloc, 1) + 8.0 / D)), 2) if traffic else None},
    }
    if pmc:
        roofline["pmc"] = {k: pmc[k] for k in ("calls_profiled", "seconds", "correction", "command")}
    if m["polyphase"]:
        A = -(-m["ntaps"] // D)
        M = tm["transform_length_M"]
        nseg = -(-(m["K_call"] + 2) // (M - A + 1))
        lg = 7 if M == 128 else 8
        flops = {"xlp_forward_kernel": 5.0 * M * lg * D * nseg, "xlp_mix_f32_kernel": 8.0 * nloc * nseg * M * D,
                 "xlp_mix_mfma_kernel": 8.0 * nloc * nseg * M * D,
                 "xlp_inverse_kernel": nloc * nseg * (5.0 * M * lg + 8.0 * (M - A + 1))}
        flops["xlp_inverse8_kernel"] = flops["xlp_inverse32_kernel"] = flops["xlp_inverse_kernel"]
        # matrix-core mix: half-precision flops the launch EXECUTES = 3 products x (32 rows x 32 columns x 16 k x 2) per k-block of
        # 8 branches, per (bin, 32 columns, pass of 16 segments = 32 rows)
        mfma_flops = 3.0 * 32 * 32 * 16 * 2 * -(-D // 8) * M * -(-nloc // 32) * -(-nseg // 16)
        binding = {"xlp_forward_kernel": "latency (a few % of the call; shared by all clients)",
                   "xlp_mix_f32_kernel": "fp32 matrix pipe (v_mfma_f32_32x32x2_f32, one per (branch, bin, 32 columns, pass): the FP32 rate of the chip)",
                   "xlp_mix_mfma_kernel": "hbm (writes the mixed spectra once, reads the operand-form branch spectra once); the products run on the "
                                          "matrix cores as two-term half-precision splits (3 v_mfma_f32_32x32x16_f16 per 8 branches)",
                   "xlp_inverse_kernel": "hbm (reads the mixed spectra, writes the outputs)"}
        binding["xlp_inverse8_kernel"] = ("hbm access pattern: with transform and phases compiled out the launch is no faster (profiles/r04_inverse8.txt); "
                                          "reads 4.7 TB/s, the output pieces of 928 bytes per (segment, client) 3.0-4.3 TB/s, and the two do not overlap")
        binding["xlp_inverse32_kernel"] = ("hbm, and how much of it a CU keeps in flight: the launch's traffic alone -- whole-line tile loads, 256-byte "
                                           "store runs, nothing else -- takes 0.82 of this kernel's time (tools/ubench_tile_copy.hip, "
                                           "profiles/r05_inverse_cut32.txt); vector ALUs 36 % busy")
        pk = {}
        trace_ms = {k: v.get("ms_per_dispatch_kernel_trace") for k, v in (pmc["per_kernel"] if pmc else {}).items()}
        for kname, ms_ev in (m["kernels_ms"] or {}).items():
            kpre = kname[:-len("_kernel")]  # (the trace prints template arguments after the name)
            b = next((v for k, v in per_kernel_bytes.items() if k.startswith(kpre) and ("mfma" in k) == ("mfma" in kname)), None)
            ms_k = next((v for k, v in trace_ms.items() if k.startswith(kpre) and ("mfma" in k) == ("mfma" in kname) and v), None) or ms_ev
            pk[kname] = {"ms": ms_k, "ms_source": "rocprofv3 --kernel-trace, this run" if ms_k is not ms_ev else "HIP events around the launch",
                         "ms_hip_events": ms_ev, "hbm_bytes": b,
                         "frac_hbm": round(b / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if b and ms_k else None,
                         "frac_fp32": round(flops[kname] / (ms_k * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4) if ms_k else None,
                         "binding": binding[kname]}
            if kname == "xlp_mix_mfma_kernel" and ms_k:
                pk[kname]["frac_fp32_note"] = "FP32-equivalent flops (8 per complex MAC) over the FP32 vector peak: what the packed-FMA kernel would need"
                pk[kname]["frac_mfma_f16"] = round(mfma_flops / (ms_k * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)
        chain = next((v for k, v in per_kernel_bytes.items() if k.startswith("xl_nco_chain")), None)
        if chain is not None:
            chain_ms = next((v for k, v in trace_ms.items() if k.startswith("xl_nco_chain") and v), None)
            pk["xl_nco_chain_kernel"] = {"ms": chain_ms, "ms_source": "rocprofv3 --kernel-trace pass over `bench.py --replay-calls` (profiler attached, "
                                         "NOT the timed region: under the profiler a chain launch runs 2-4 % longer than in the timed run, so "
                                         "ms x launches per step may exceed ms_per_step; ms_per_call_timed_region below is the timed region's own bound)",
                                         "ms_per_call_profiled": round(chain_ms / 4.0, 4) if chain_ms else None,
                                         "ms_per_call_timed_region_upper_bound": round(period_ms, 4),
                                         "hbm_bytes": chain, "binding": "a dependent float32 recurrence on a side stream (reserved CUs), "
                                         "concurrent with the three launches; bounds the engine below ~1500 clients",
                                         "note": "bytes per CALL, ms per LAUNCH (one launch tabulates the phase tables of four calls); in the timed region a "
                                                 "call cannot be shorter than a quarter of a chain launch, so that launch is at most 4 x call_period_ms"}
        roofline["kernel"] = ("xlp_forward_kernel + " + ("xlp_mix_mfma_kernel" if "mix=mfma" in m["plan"] else "xlp_mix_f32_kernel") + " + " + next((k for k in (m["kernels_ms"] or {}) if k.startswith("xlp_inverse")), "xlp_inverse_kernel") + ": the three launches of one call on the polyphase "
                              "overlap-save path (the next call's NCO phase recurrence runs beside them on a side stream)")
        roofline["per_kernel"] = pk
        roofline["kernel_short"] = "+".join(k.replace("_kernel", "") for k in (m["kernels_ms"] or {})) + " (the 3 launches of one call)"
        # what bounds the STEP at this client count: the next call's float32 NCO phase recurrence (the reference's own p *= incr chain,
        # src/xlating.c:70-73: 3121 dependent steps per block and client) runs beside the launches on reserved CUs; one chain launch
        # tabulates CHAIN_CALLS calls.  frac_of_recurrence_floor = its share of the call period: ~1 = the step IS the recurrence
        # (<= ~1500 clients); well below 1 = the launches (HBM) bound the step (2048 / 4096 clients: configs.clients_*)
        ck = pk.get("xl_nco_chain_kernel") or {}
        if ck.get("ms_per_call_profiled"):
            roofline["chain_ms_per_call"] = ck["ms_per_call_profiled"]
            roofline["frac_of_recurrence_floor"] = round(min(1.0, ck["ms_per_call_profiled"] / period_ms), 4) if period_ms > 0 else None
            roofline["step_bound"] = "nco_recurrence" if roofline["frac_of_recurrence_floor"] and roofline["frac_of_recurrence_floor"] >= 0.9 else "hbm"
        roofline["per_kernel_note"] = ("ms: the kernel's own duration from a rocprofv3 --kernel-trace pass of this run over `bench.py --replay-calls` -- a "
                                       "separate, profiled pass, not the timed region (ms_hip_events: event pairs around "
                                       "each launch, 16 extra calls after the timed region -- they include the event records and forbid the overlap "
                                       "of one launch's tail with the next one's head, so their sum exceeds kernel_ms)")
        roofline["design_traffic"] = dict(tm, note="bytes the path moves through HBM per call by design (xl_polyphase.h); compare with 'traffic'")
    else:
        kname = next((k for k in per_kernel_bytes if k.startswith("xl_fir_kernel")), "xl_fir_kernel")
        roofline["kernel"] = (f"xl_fir_kernel<H,{1 if args.mode == 'optimized' else 0},wide> (H = register-tile height chosen by the "
                              "engine; one launch per call: history roll + FIR + next call's NCO phase table)")
        roofline["kernel_short"] = "xl_fir (one launch per call)"
        roofline["per_kernel"] = {kname: {"ms": round(m["call_ms_avg"], 4), "hbm_bytes": per_kernel_bytes.get(kname),
                                          "frac_hbm": round(ach / HBM_PEAK_GBS, 4), "frac_fp32": round(ach_tf / FP32_PEAK_TFLOPS, 4),
                                          "binding": "fp32 vector issue (native: 4 packed unfused ops per complex MAC -> ceiling 0.5; optimized: 2 packed FMAs)"}}
        roofline["fp32"] = {"achieved": round(ach_tf, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(ach_tf / FP32_PEAK_TFLOPS, 4), "flop_per_unit": round(fpu, 2)}

    # the numbers of this workload whose every product is a float32 product, next to the headline (whose mix launch multiplies two-half
    # operands): bit-exact native, optimized through the direct FMA kernel, optimized polyphase with float32 operands in the mix launch
    def pick(prefix):
        k = next((k for k in variants if k.startswith(prefix)), None)
        return None if k is None else {kk: variants[k].get(kk) for kk in ("value", "us_per_block", "parity_spot") if variants[k].get(kk) is not None}
    all_f32 = {"native (bit-exact, direct kernel: the server default NATIVE_CF32)": None if native is None else
               {"value": native["value"], "us_per_block": native["us_per_block"], "parity_spot": native.get("parity_spot")},
               "optimized, direct FMA kernel": pick(f"process_{args.mode}_cu8_cf32 through the direct FIR kernel"),
               "optimized, polyphase with the float32 matrix-core mix": pick("polyphase, float32 matrix-core mix")}
    # N > 1: both ways to use the GPUs, fully populated (the headline is one of them)
    multi_gpu = None
    if world > 1:
        head = {"value": round(value, 1), "ms_per_step": round(m["seconds"] / args.steps * 1e3, 4), "clients_total": total_clients,
                "clients_per_gpu": nloc, "per_rank_seconds": m["rank_seconds"], "feed_timing": m["feed_timing"],
                "rccl_comm_count": m["rccl_comm_count"], "scaling": args.scaling}
        other = "weak" if args.scaling == "strong" else "strong"
        ko = next((k for k in variants if k.startswith(other + " scaling")), None)
        multi_gpu = {args.scaling: head, other: (dict(variants[ko], scaling=other) if ko else None),
                     "note": "strong = BASELINE configs[3] read literally (1024 clients in total); weak = 1024 clients per GPU; per_rank_seconds: "
                             "every rank's wall time of each timed repeat (the value uses the max); feed_timing: rank 0's broadcasts"}
    rep_ms = [round(sec / args.steps * 1e3, 4) for sec in m["repeat_seconds"]]
    out = {
        "metric": "input IQ Msamples/s processed (all clients), 2.016 Msps->48 kHz xlating FIR",
        "value": round(value, 1),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(m["seconds"] / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": ("f32 (mix products: 2xf16 split, f32 accumulate)" if (m["polyphase"] and "mix=mfma" in m["plan"]) else "f32"),
        "dtype_note": ("float32 in, float32 out, float32 accumulation everywhere; on the polyphase path the per-client sums (mix launch) "
                       "multiply float32 operands carried as two half-precision terms each (products exact in float32, 2^-22 per product "
                       "dropped: as accurate as the float32 FMA chain, tests/test_mix_split_model.py) -- not a reduced-precision run: "
                       "parity_spot holds every client to the 1e-5 bar against the float32 reference.  The all-float32 and bit-exact "
                       "numbers of the same workload are in `all_f32`") if (m["polyphase"] and "mix=mfma" in m["plan"]) else "float32 throughout",
        "data": "synthetic" if cuda else "cpu-plumbing-test",
        "repeats": {"n": len(rep_ms), "ms_per_step": rep_ms, "min": min(rep_ms), "median": sorted(rep_ms)[len(rep_ms) // 2], "max": max(rep_ms),
                    "value_is": "the median repeat; each repeat = exactly `steps` steps between barrier + synchronize, max over ranks",
                    "values": [round(total_clients * S * args.steps * m["blocks_per_step"] / sec / 1e6, 1) for sec in m["repeat_seconds"]]},
        "config": {
            "workload": f"{total_clients} clients x 48 kHz off one 2.016 Msps cu8 stream ({nloc} on this GPU), {BLOCK_BYTES}-byte blocks, "
                        f"D={D}, {m['ntaps']} taps (lpf_cutoff_rate={args.lpf_cutoff_rate}), process_{args.mode}_cu8_cf32 semantics, "
                        f"{GROUP} blocks per engine call (BASELINE configs[3]; N=1: the >=1000-client single-GPU target; "
                        + ("weak scaling: 1024 clients per GPU -- configs[3]'s 1024 in total is variants['strong scaling ...']" if args.scaling == "weak" else
                           "strong scaling: the total is fixed") + ")",
            "step": f"{m['blocks_per_step']} consecutive blocks = {calls_per_step} calls = {m['blocks_per_step'] * S} stream samples per client",
            "clients_total": total_clients, "clients_per_gpu": nloc, "ntaps": m["ntaps"], "mode": args.mode, "block_samples": S, "blocks_per_call": GROUP,
            "mix_products": ("2xf16 split operands on the matrix cores, f32 accumulate (every client <= 1e-5 vs the f32 reference: parity_spot; "
                             "exact-f32 number: configs.all_f32)" if (m["polyphase"] and "mix=mfma" in m["plan"]) else "f32"),
            "outputs_per_client_per_call": m["K_call"], "us_per_block": round(m["seconds"] / (args.steps * m["blocks_per_step"]) * 1e6, 3),
            "parallelism": (f"clients sharded c%{world}; one RCCL broadcast per {GROUP} raw IQ blocks ({GROUP * BLOCK_BYTES} bytes) "
                            "on a separate stream (overlaps the previous call's filtering), no other collective") if world > 1 else "single GPU",
            "rccl_ranks": rccl_ranks, "rank0_device": (f"cuda:{local_rank}" if cuda else "cpu"), "feed": m["feed"],
        },
        "roofline": roofline,
        "parity_spot": m["parity_spot"],
        "plan": m["plan"],
        "native": native,
        "all_f32": all_f32,
        "multi_gpu": multi_gpu,
        "variants": variants,
        "expected_scaling": {"strong": EXPECTED_STRONG, "weak": EXPECTED_WEAK,
                             "note": "what DESIGN.md section 8 (history: DESIGN_HISTORY.md section 7) predicts for --gpus 1/2/4/8, to judge a measured curve against"},
        "device": xl.device_info() if cuda else "none (plumbing test)",
    }
    if world == 1 and not args.no_cpu_baseline and cuda:
        out["cpu_baseline"] = cpu_baseline(args.lpf_cutoff_rate, args.cpu_seconds)
    # the full record goes to a file; stdout gets ONE compact line (<= 4 KB) the driver can parse
    full_path = args.full_json or (FULL_JSON_DEFAULT if cuda else os.path.join("/tmp", f"xl_bench_full_{os.getpid()}.json"))
    out["full_record"] = os.path.relpath(full_path, ROOT) if full_path.startswith(ROOT) else full_path
    try:
        os.makedirs(os.path.dirname(full_path), exist_ok=True)
        with open(full_path, "w") as fh:
            json.dump(out, fh, indent=1)
        gout = os.path.join(ROOT, "gpurun_out")  # (on a gpurun box only this directory travels back)
        if cuda and os.path.isdir(gout):
            with open(os.path.join(gout, "bench_last_full.json"), "w") as fh:
                json.dump(out, fh, indent=1)
    except OSError as e:
        out["full_record"] = f"not written ({e})"
    print(f"bench.py: full record ({len(json.dumps(out))} bytes) -> {out['full_record']}", file=sys.stderr, flush=True)
    print(compact_line(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
