#!/usr/bin/env python3
"""bench.py -- throughput of the xlating-FIR hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]            (N > 1: launched by torch.distributed.run)

A "step" = one pass of the hot path over one IQ block for every client of every GPU:
    N > 1: rank 0's block is broadcast over RCCL/xGMI (the path's only exchange step), then each rank runs its
    own clients on it -- no further communication (clients are embarrassingly parallel).
Workload (config.workload): 1024 concurrent 48 kHz clients PER GPU off one 2.016 Msps cu8 stream, server-default
262144-byte blocks (131072 complex samples), D=42, low-pass designed with the server default lpf_cutoff_rate=5
-> 505 taps (the BASELINE "~84-tap" figure is not reachable with the reference's designer; the
lpf_cutoff_rate=1 -> 101-tap variant is measured too and reported under "variants").  Weak scaling: per-GPU work
is fixed as N grows.  Inputs are synthetic (xorshift bytes), already resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0): value = input IQ Msamples/s summed over all clients and GPUs.
"roofline":     HBM-read roofline of the block's launches, per-client-read model of SURVEY 8(d): algorithmic bytes per
                block = clients x samples x (2 B in + 8/D B out), divided by the mean duration of the block's
                launches measured with HIP events on the launch stream inside the timed region.  The optimized
                variant of this workload runs the polyphase overlap-save kernels (three launches per block:
                forward / mix / inverse, xl_polyphase.hip) -- HBM-bound; their separate durations come from a
                short extra pass after the timed region ("kernels_ms").  The direct FIR kernel (what the native
                variant and short filters use; 96 flop per (client, sample) at 505 taps: FP32-bound) is reported
                under "variants" with its FP32 fraction.
"cpu_baseline": the reference itself (oracle/_ref, unmodified sources, -O3 -ffast-math AVX2) -- or the repo's CPU
                restatement when that build is absent -- timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
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
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: FP32 vector == FP32 MFMA peak
TIMING_STRIDE = 4               # HIP events bracket every 4th block of the timed region (an event pair costs ~6 us of
                                # stream time: measured 65.7 us per block with a pair per block, 59.9 us with one per 4)


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
    """The path's only exchange step: rank 0's raw IQ block -> every rank (RCCL over xGMI on GPUs, gloo in the
    CPU tests).  `recv` is each rank's receive buffer; returns the tensor holding the block on this rank."""
    if rank == 0:
        recv.copy_(src_block, non_blocking=True)
    dist.broadcast(recv, src=0)
    return recv


def reduce_max_seconds(dist, torch, dt, device):
    """bench contract: the step time is the MAX over ranks."""
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def make_blocks(nblocks, seed):
    import siggen

    return [siggen.xs_u8(seed + k, BLOCK_BYTES) for k in range(nblocks)]


def cpu_baseline(ntaps_rate, seconds=12.0):
    """Reference (oracle/_ref/libref_fast.so) -- or the repo's CPU restatement when that build is absent -- on the
    host cores, via the native pthread driver oracle/cpu_bench: one thread per client, each with its own filter
    over the same block (the reference's dsp_worker model, src/dsp_worker.c:41-88).  Bounded sample: `seconds` of
    wall time split between a 1-thread and an all-cores run."""
    import subprocess

    odir = os.path.join(ROOT, "oracle")
    drv = os.path.join(odir, "cpu_bench")
    flags = open("/proc/cpuinfo").read()
    model = next((l.split(":", 1)[1].strip() for l in flags.splitlines() if l.startswith("model name")), "unknown")
    flat = " " + flags.replace("\n", " ") + " "
    ref = os.path.join(odir, "_ref", "libref_fast.so")
    if os.path.exists(ref) and " avx2 " in flat and " fma " in flat:
        lib, api, variant, kind = ref, "ref", "optimized", "reference"
        what = "unmodified src/xlating.c process_optimized_cu8_cf32 (hand-written AVX path; gcc -O3 -ffast-math -mavx2 -mfma)"
    else:
        lib, api, variant, kind = os.path.join(odir, "liboracle.so"), "orc", "native", "port"
        what = "oracle/xlating_oracle.c scalar restatement (gcc -O2, canonical order)"
    cores = os.cpu_count() or 1

    def run(threads, secs):
        r = subprocess.run([drv, lib, api, variant, str(threads), str(secs), str(FS), str(RATE), str(RATE // ntaps_rate),
                            str(BLOCK_BYTES)], capture_output=True, text=True, timeout=secs * 3 + 60)
        if r.returncode != 0:
            raise RuntimeError(f"cpu_bench failed: {r.stderr[-500:]}")
        return json.loads(r.stdout.strip().splitlines()[-1])

    one = run(1, max(2.0, seconds * 0.3))
    allc = run(cores, max(3.0, seconds * 0.7))
    return {
        "value": round(allc["msps"], 1), "unit": "Msamples/s", "cores": cores, "kind": kind,
        "single_thread_value": round(one["msps"], 1),
        "sample": f"{what}; lpf_cutoff_rate={ntaps_rate} -> {allc['ntaps']} taps, D={D}, {BLOCK_BYTES}-byte cu8 blocks, one "
                  f"filter per thread over a shared block; {allc['calls']} calls on {cores} threads in {allc['seconds']:.1f} s "
                  f"wall (+ {one['calls']} calls on 1 thread in {one['seconds']:.1f} s); host CPU: {model}",
    }


FEED_GROUP = 8  # blocks per RCCL broadcast (SURVEY 8(d) config 4 names the 8-block super-block): the cross-stream event
                # pair that orders a broadcast against the filtering costs ~10 us of stream time (tools/feed_overhead.py:
                # 58.7 -> 68.8 us per block with one pair per block), a sixth of a block's launches


class BlockFeeder:
    """Delivers block k to this rank: N = 1 -> a resident device buffer; N > 1 -> rank 0's blocks broadcast over
    RCCL/xGMI, FEED_GROUP blocks per broadcast, into one of two alternating receive buffers ON A SEPARATE STREAM, so
    that the broadcast of group g+1 overlaps the filtering of group g (events order buffer reuse: the broadcast into a
    buffer waits until the launches that read it two groups ago have been passed by the compute stream)."""

    def __init__(self, torch, dist, rank, world, dev_blocks, group=FEED_GROUP):
        self.torch, self.dist, self.rank, self.world, self.blocks = torch, dist, rank, world, dev_blocks
        self.group = group
        self.issued = 0  # groups issued
        if world > 1:
            self.recv = [torch.empty(group * BLOCK_BYTES, dtype=torch.uint8, device="cuda") for _ in range(2)]
            self.comm = torch.cuda.Stream()
            self.ready = [torch.cuda.Event() for _ in range(2)]
            self.free = [torch.cuda.Event() for _ in range(2)]
            self.free_valid = [False, False]

    def _issue(self, g):
        torch = self.torch
        i = g % 2
        if self.free_valid[i]:
            self.comm.wait_event(self.free[i])
        with torch.cuda.stream(self.comm):
            if self.rank == 0:
                for j in range(self.group):
                    src = self.blocks[(g * self.group + j) % len(self.blocks)]
                    self.recv[i][j * BLOCK_BYTES:(j + 1) * BLOCK_BYTES].copy_(src, non_blocking=True)
            self.dist.broadcast(self.recv[i], src=0)  # the path's only exchange step
            self.ready[i].record(self.comm)
        self.issued = g + 1

    def get(self, k, stream):
        """Device pointer of block k, valid on `stream`; call consumed(k, stream) after enqueuing its consumer."""
        if self.world == 1:
            return self.blocks[k % len(self.blocks)].data_ptr()
        g, j = divmod(k, self.group)
        while self.issued <= g + 1:  # keep one broadcast in flight ahead of the consumer
            self._issue(self.issued)
        if j == 0:
            stream.wait_event(self.ready[g % 2])
        return self.recv[g % 2].data_ptr() + j * BLOCK_BYTES

    def consumed(self, k, stream):
        if self.world > 1 and k % self.group == self.group - 1:
            g = k // self.group
            self.free[g % 2].record(stream)
            self.free_valid[g % 2] = True


def run_workload(xl, torch, dist, args, rank, world, ntaps_rate, steps, warmup, dev_blocks, mode=None, poly=None):
    """Build this rank's engine with its shard of clients and time `steps` blocks.  Returns dict of measurements.
    poly: None = the engine's own choice of arithmetic path for the optimized variant; 0 = direct FIR kernels only
    (the engine reads the XL_EXP_POLY tuning switch when it is created)."""
    code, taps = xl.create_low_pass_filter(1.0, FS, RATE // 2, RATE // ntaps_rate)
    assert code == 0
    total_clients = args.clients_per_gpu * world
    mine = shard_clients(total_clients, world, rank)
    saved = os.environ.get("XL_EXP_POLY")
    if poly is not None:
        os.environ["XL_EXP_POLY"] = str(poly)
    eng = xl.BatchEngine(FS, "cu8", BLOCK_BYTES, device=torch.cuda.current_device())
    if poly is not None:
        if saved is None:
            del os.environ["XL_EXP_POLY"]
        else:
            os.environ["XL_EXP_POLY"] = saved
    for c in mine:
        eng.add_client(D, taps, client_center_freq(c))
    stream = torch.cuda.current_stream()
    feeder = BlockFeeder(torch, dist, rank, world, dev_blocks)

    def step(k):
        ptr = feeder.get(k, stream)
        eng.process_device(ptr, BLOCK_BYTES, mode or args.mode, stream.cuda_stream)
        feeder.consumed(k, stream)

    for k in range(warmup):
        step(k)
    eng.sync()
    torch.cuda.synchronize()
    eng.timing_stride(TIMING_STRIDE)
    eng.timing(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(warmup + k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nt, fir_ms, nco_ms = eng.timing_read(reset=True)
    eng.timing(False)
    if world > 1:
        dt = reduce_max_seconds(dist, torch, dt, "cuda")
    klen = eng.output_len(0)
    plan = eng.describe()
    use_mode = mode or args.mode
    polyphase = use_mode == "optimized" and "polyphase: none" not in plan
    kernels_ms = None
    if polyphase:  # separate durations of the three launches: a short extra pass, OUTSIDE the timed region
        eng.timing_stride(1)
        eng.timing(2)
        extra = 40
        for k in range(extra):
            step(warmup + steps + k)
        torch.cuda.synchronize()
        n3, ms3 = eng.timing_polyphase(reset=True)
        eng.timing(False)
        if n3 > 0:
            kernels_ms = {"xlp_forward_kernel": round(ms3[0] / n3, 4), "xlp_mix_kernel": round(ms3[1] / n3, 4),
                          "xlp_inverse_kernel": round(ms3[2] / n3, 4)}
    eng.close()
    return {"ntaps": int(taps.size), "seconds": dt, "fir_ms_avg": fir_ms / max(nt, 1), "nco_ms_avg": nco_ms / max(nt, 1),
            "timed_launches": nt, "clients_this_rank": len(mine), "total_clients": total_clients, "K": int(klen),
            "plan": plan, "polyphase": polyphase, "kernels_ms": kernels_ms}


def polyphase_traffic_model(nclients, K, ntaps, M=128):
    """HBM bytes one block moves by design on the polyphase path (xl_polyphase.h), per GPU: branch spectra R read
    once (8 D M bytes per client), mixed spectra Y written and read back (8 M bytes per client and segment), outputs
    written, NCO phase table (every 16th phase) written and read; the shared spectra X and the raw block are noise."""
    A = -(-ntaps // D)
    V = M - A + 1
    nseg = -(-K // V)
    dpad = -(-D // 7) * 7
    per_client = 8 * dpad * M + 2 * 8 * M * nseg + 8 * K + 2 * 8 * (K // 16)
    return {"transform_length_M": M, "bytes_per_block": int(nclients * per_client), "bytes_per_client": int(per_client),
            "R_branch_spectra": 8 * dpad * M, "Y_mixed_spectra_write_plus_read": 2 * 8 * M * nseg,
            "out": 8 * K, "phase_table_write_plus_read": 2 * 8 * (K // 16)}


def summarize(m, steps, world):
    """value and roofline figures from the timed region of `m` (HIP-event duration of the FIR launches in it)."""
    units_per_step = m["total_clients"] * S
    value = units_per_step * steps / m["seconds"] / 1e6
    per_gpu_units = m["clients_this_rank"] * S
    bpu = algorithmic_bytes_per_unit(D)
    fpu = flops_per_unit(m["ntaps"], D)
    fir_s = m["fir_ms_avg"] * 1e-3
    ach_gbs = per_gpu_units * bpu / fir_s / 1e9 if fir_s > 0 else 0.0
    ach_tf = per_gpu_units * fpu / fir_s / 1e12 if fir_s > 0 else 0.0
    return value, ach_gbs, ach_tf, bpu, fpu


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--clients-per-gpu", type=int, default=1024)
    ap.add_argument("--mode", default="optimized", choices=["native", "optimized"])
    ap.add_argument("--lpf-cutoff-rate", type=int, default=5, help="server config lpf_cutoff_rate: 5 -> 505 taps, 1 -> 101")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with python -m torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device (there is no CPU path to time)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import sdr_server_amd as xl

    blocks = make_blocks(8, 0x5DEECE66D)
    dev_blocks = [torch.from_numpy(b).cuda() for b in blocks] if (rank == 0 or world == 1) else []

    m = run_workload(xl, torch, dist, args, rank, world, args.lpf_cutoff_rate, args.steps, args.warmup, dev_blocks)
    value, ach_gbs, ach_tf, bpu, fpu = summarize(m, args.steps, world)

    variants = {}
    if not args.no_variants:
        other_rate = 1 if args.lpf_cutoff_rate != 1 else 5
        vs = max(20, args.steps // 2)
        mv = run_workload(xl, torch, dist, args, rank, world, other_rate, vs, min(args.warmup, 5), dev_blocks)
        v2, g2, t2, _, f2 = summarize(mv, vs, world)
        variants[f"lpf_cutoff_rate={other_rate} ({mv['ntaps']} taps)"] = {
            "value": round(v2, 1), "ms_per_step": round(mv["seconds"] / vs * 1e3, 4),
            "path": "polyphase" if mv["polyphase"] else "direct FIR kernel",
            "roofline_hbm_frac": round(g2 / HBM_PEAK_GBS, 4), "achieved_GBs": round(g2, 1),
            "fp32_frac": None if mv["polyphase"] else round(t2 / FP32_PEAK_TFLOPS, 4),
            "achieved_TFLOPs": None if mv["polyphase"] else round(t2, 2),
            "kernel_ms": round(mv["fir_ms_avg"], 4),
            "flop_per_unit": round(f2, 2)}
        # the other arithmetic variant on the headline workload (native = bit-exact reference arithmetic,
        # the reference's default cpu_optimization: always the direct FIR kernel)
        other_mode = "native" if args.mode == "optimized" else "optimized"
        mn = run_workload(xl, torch, dist, args, rank, world, args.lpf_cutoff_rate, vs, min(args.warmup, 5), dev_blocks,
                          mode=other_mode)
        v3, g3, t3, _, _ = summarize(mn, vs, world)
        variants[f"process_{other_mode}_cu8_cf32 semantics ({mn['ntaps']} taps)"] = {
            "value": round(v3, 1), "ms_per_step": round(mn["seconds"] / vs * 1e3, 4), "kernel_ms": round(mn["fir_ms_avg"], 4),
            "path": "polyphase" if mn["polyphase"] else "direct FIR kernel",
            "roofline_hbm_frac": round(g3 / HBM_PEAK_GBS, 4),
            "achieved_TFLOPs": None if mn["polyphase"] else round(t3, 2),
            "fp32_frac": None if mn["polyphase"] else round(t3 / FP32_PEAK_TFLOPS, 4)}
        if m["polyphase"]:  # the same workload through the direct FIR kernels (tuning switch): the FP32-bound design
            md = run_workload(xl, torch, dist, args, rank, world, args.lpf_cutoff_rate, vs, min(args.warmup, 5), dev_blocks,
                              poly=0)
            v4, g4, t4, _, f4 = summarize(md, vs, world)
            variants[f"process_{args.mode}_cu8_cf32 through the direct FIR kernel ({md['ntaps']} taps)"] = {
                "value": round(v4, 1), "ms_per_step": round(md["seconds"] / vs * 1e3, 4), "kernel_ms": round(md["fir_ms_avg"], 4),
                "path": "direct FIR kernel", "roofline_hbm_frac": round(g4 / HBM_PEAK_GBS, 4),
                "achieved_TFLOPs": round(t4, 2), "fp32_frac": round(t4 / FP32_PEAK_TFLOPS, 4), "flop_per_unit": round(f4, 2),
                "note": "FP32-bound: 96 flop per (client, sample) caps the HBM fraction at "
                        f"{min(1.0, FP32_PEAK_TFLOPS * 1e12 / f4 * bpu / (HBM_PEAK_GBS * 1e9)):.3f}"}

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            traffic = json.load(open(pmc_path)).get("hbm_bytes_per_block_polyphase" if m["polyphase"] else "hbm_bytes_per_launch")
        except Exception:
            traffic = None

    kernel_s = m["fir_ms_avg"] * 1e-3
    nloc = m["clients_this_rank"]
    roofline = {
        "bound": "hbm", "achieved": round(ach_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(ach_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
        "kernel_ms": round(m["fir_ms_avg"], 4),
        "kernel_ms_note": f"mean HIP-event duration of one block's launches, every {TIMING_STRIDE}th block of the timed region, on the launch stream",
        "bytes_per_unit": round(bpu, 4), "units_per_launch": nloc * S,
        "model": "per-client-read (SURVEY 8(d)): 2 B in + 8/D B out per (client, input sample)",
        "shared_read_model": {"bytes_per_unit": round(2.0 / nloc + 8.0 / D, 5),
                              "achieved_GBs": round(nloc * S * (2.0 / nloc + 8.0 / D) / kernel_s / 1e9, 1),
                              "note": "the block is read once per GPU and shared through L2/LDS: 2/N + 8/D B per unit (SURVEY 8(d))"},
    }
    if m["polyphase"]:
        import re
        mm = re.search(r"polyphase: cls0 .*? M(\d+)", m["plan"])
        tm = polyphase_traffic_model(nloc, m["K"], m["ntaps"], int(mm.group(1)) if mm else 256)
        roofline["kernel"] = ("xlp_forward_kernel + xlp_mix_kernel (dominant) + xlp_inverse_kernel: the three launches of one "
                              "block on the polyphase overlap-save path (each also carries a slice of the next block's NCO "
                              "phase recurrence; the forward launch rolls the raw history)")
        roofline["kernels_ms"] = m["kernels_ms"]
        roofline["kernels_ms_note"] = "separate HIP-event durations of the three launches, 40 extra blocks after the timed region"
        roofline["design_traffic"] = dict(tm, achieved_GBs=round(tm["bytes_per_block"] / kernel_s / 1e9, 1),
                                          frac_of_peak=round(tm["bytes_per_block"] / kernel_s / 1e9 / HBM_PEAK_GBS, 4),
                                          note="bytes the path moves through HBM per block by design (xl_polyphase.h); compare with 'traffic'")
        roofline["direct_equivalent_TFLOPs"] = round(ach_tf, 2)
        roofline["direct_equivalent_note"] = (f"what the reference's direct {m['ntaps']}-tap dot product would need for this rate "
                                               f"({fpu:.1f} flop per unit; FP32 vector peak {FP32_PEAK_TFLOPS} TFLOP/s) -- the polyphase path "
                                               "does ~10x fewer flops, which is why it can pass the FP32 ceiling of the direct kernel")
    else:
        roofline["kernel"] = (f"xl_fir_kernel<H,{1 if args.mode == 'optimized' else 0},wide> (H = register-tile height chosen by the "
                              "engine; one launch per block: history roll + FIR + next block's NCO phase table)")
        roofline["fp32"] = {"achieved": round(ach_tf, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(ach_tf / FP32_PEAK_TFLOPS, 4), "flop_per_unit": round(fpu, 2),
                            "note": "binding ceiling at this tap count (SURVEY H2): HBM frac cannot exceed "
                                    f"{min(1.0, FP32_PEAK_TFLOPS * 1e12 / fpu * bpu / (HBM_PEAK_GBS * 1e9)):.3f}"}
        roofline["nco_table_kernel_ms"] = round(m["nco_ms_avg"], 4) if m["nco_ms_avg"] > 0 else "fused into the FIR launch"

    out = {
        "metric": "input IQ Msamples/s processed (all clients), 2.016 Msps->48 kHz xlating FIR",
        "value": round(value, 1),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(m["seconds"] / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.clients_per_gpu} clients/GPU x 48 kHz off one 2.016 Msps cu8 stream, {BLOCK_BYTES}-byte blocks, "
                        f"D={D}, {m['ntaps']} taps (lpf_cutoff_rate={args.lpf_cutoff_rate}), process_{args.mode}_cu8_cf32 semantics "
                        f"(BASELINE configs[3] per-GPU share x8 = the 1-GPU >=1000-client target)",
            "clients_total": m["total_clients"], "block_samples": S, "outputs_per_client_per_block": m["K"],
            "parallelism": (f"clients sharded c%{world}; one RCCL broadcast per {FEED_GROUP} raw IQ blocks ({FEED_GROUP * BLOCK_BYTES} bytes) "
                            "on a separate stream (overlaps the previous group's filtering), no other collective") if world > 1 else "single GPU",
        },
        "roofline": roofline,
        "plan": m["plan"],
        "variants": variants,
        "device": xl.device_info(),
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.lpf_cutoff_rate, args.cpu_seconds)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
