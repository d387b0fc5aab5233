#!/usr/bin/env python3
"""tools/measure_configs.py -- numbers for every BASELINE.json config on one MI355X (parity for them is in tests/):
 [1] 1 client, 2.016 Msps -> 48 kHz through the drop-in process_* API (latency-bound: us per 262144-byte block)
 [2] 64 clients at mixed 48/96 kHz sharing one block (batch engine, device-resident input)
 [3] per-GPU share of the 8-GPU 1024-client config (128 clients) and the 1-GPU 1024-client target
 [4] cf32 input at 10 Msps, D=100, 257 taps, 256 clients (HBM-roofline style config)
 plus the PCIe-inclusive host path (process_host + fetch of every client's output).
Prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import siggen  # noqa: E402
import sdr_server_amd as xl  # noqa: E402

FS = 2016000
out = {"device": xl.device_info()}


def lpf(fs, cut, tw):
    return xl.create_low_pass_filter(1.0, fs, cut, tw)[1]


# ---- [1] drop-in single filter: measured by a torch-free child process (tools/measure_dropin.py says why)
import subprocess  # noqa: E402

_r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "measure_dropin.py")], capture_output=True, text=True, timeout=300)
out["config1_single_client_dropin_process_cu8_cf32"] = json.loads(_r.stdout.strip().splitlines()[-1]) if _r.returncode == 0 else {"error": _r.stderr[-300:]}
# ... and from C, the way dsp_worker.c:49-86 calls the filter: 10 000 consecutive calls, no Python in the process
# (tests/c/dropin_latency.c; the 1-35 ms outliers of round 2's numbers were CPython's garbage collector in the measuring
# process: tools/dropin_python_stall.py)
_lat = os.path.join(ROOT, "sdr-server_amd", "build", "dropin_latency")
if os.path.exists(_lat):
    out["config1_dropin_latency_from_C_10000_calls"] = {}
    for _v in ("native", "optimized"):
        _r = subprocess.run([_lat, _v, "10000"], capture_output=True, text=True, timeout=300)
        if _r.returncode == 0:
            _j = json.loads(_r.stdout.strip().splitlines()[-1])
            out["config1_dropin_latency_from_C_10000_calls"][_v] = {k: _j[k] for k in ("mean_us", "median_us", "p99_us", "p999_us", "max_us", "calls_over_1ms")}


GROUP = 8  # blocks per engine call on the device-resident path (bench.py's super-block)


def run_batch(fs, fmt, nbytes_per_block, clients, blocks, steps=60, variant="optimized", host=False, group=GROUP):
    """-> (seconds per BLOCK, launch milliseconds per BLOCK)"""
    g = 1 if host else group
    eng = xl.BatchEngine(fs, fmt, nbytes_per_block, group_blocks=g)
    for D, taps, fc in clients:
        eng.add_client(D, taps, fc)
    nelem = blocks[0].size
    dev = None if host else torch.from_numpy(np.concatenate([blocks[k % len(blocks)] for k in range(g)])).cuda()

    def step(k):
        if host:
            eng.process_host(blocks[k % len(blocks)], variant)
            eng.fetch()
        else:
            eng.process_device_group(dev.data_ptr(), nelem, g, variant, "engine")

    for k in range(8):
        step(k)
    eng.sync()
    torch.cuda.synchronize()
    eng.timing_stride(2)
    eng.timing(True)
    calls = max(8, steps // g)
    t0 = time.perf_counter()
    for k in range(calls):
        step(k)
    eng.sync()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (calls * g)
    nt, fir, _ = eng.timing_read()
    eng.close()
    return dt, fir / max(nt, 1) / g


t48, t96 = lpf(FS, 24000, 9600), lpf(FS, 48000, 19200)
blocks = [siggen.xs_u8(100 + k, 262144) for k in range(4)]
S = 131072
# ---- [2] 64 mixed clients
cl = [((42, t48) if c % 2 == 0 else (21, t96)) + (-900000 + c * 28000,) for c in range(64)]
dt, fir = run_batch(FS, "cu8", 262144, cl, blocks, steps=320)
out["config2_64_clients_mixed_48k_96k"] = {"us_per_block": round(dt * 1e6, 2), "launches_us_per_block": round(fir * 1e3, 2),
                                           "Msps_all_clients": round(64 * S / dt / 1e6, 0)}
dt1, fir1 = run_batch(FS, "cu8", 262144, cl, blocks, group=1)
out["config2_64_clients_mixed_48k_96k_one_block_per_call"] = {"us_per_block": round(dt1 * 1e6, 2), "Msps_all_clients": round(64 * S / dt1 / 1e6, 0)}
out["blocks_per_call_device_paths"] = GROUP
# ---- [3] per-GPU shares of the 1024-client config (8 / 4 / 2 GPUs: 128 / 256 / 512 clients) and the 1-GPU target
for n in (128, 256, 512, 1024, 2048, 4096):
    cl = [(42, t48, -984000 + 1920 * (c % 1024) + 240 * (c // 1024)) for c in range(n)]
    dt, fir = run_batch(FS, "cu8", 262144, cl, blocks, steps=320)
    out[f"config3_{n}_clients_48k_505taps"] = {"us_per_block": round(dt * 1e6, 2), "launches_us_per_block": round(fir * 1e3, 2),
                                               "Msps_all_clients": round(n * S / dt / 1e6, 0)}
for n in (128, 1024):
    cl = [(42, t48, -984000 + 1920 * c) for c in range(n)]
    dt, fir = run_batch(FS, "cu8", 262144, cl, blocks, steps=160, variant="native")
    out[f"config3_{n}_clients_48k_505taps_native"] = {"us_per_block": round(dt * 1e6, 2), "launches_us_per_block": round(fir * 1e3, 2),
                                                      "Msps_all_clients": round(n * S / dt / 1e6, 0)}
# ---- PCIe-inclusive host path
for n in (64, 1024):
    cl = [(42, t48, -984000 + 1920 * c) for c in range(n)]
    dt, fir = run_batch(FS, "cu8", 262144, cl, blocks, steps=30, host=True)
    out[f"host_path_pcie_inclusive_{n}_clients"] = {"ms_per_block": round(dt * 1e3, 3), "Msps_all_clients": round(n * S / dt / 1e6, 0),
                                                    "note": "process_host (H2D of the block) + fetch (D2H of every client's output), synchronous"}
# ---- [4] cf32 10 Msps, D=100, 257 taps
taps = siggen.hamming_sinc(257, 0.004)
fblocks = [(siggen.xs_s16(300 + k, 2 * S).astype(np.float32) / np.float32(32768)) for k in range(3)]
for n in (64, 256, 1024):
    cl = [(100, taps, -4000000 + (8000000 // n) * c) for c in range(n)]
    dt, fir = run_batch(10000000, "cf32", 2 * S, cl, fblocks, steps=640)
    bpu = 8 + 8 / 100
    out[f"config4_cf32_10Msps_D100_257taps_{n}_clients"] = {
        "us_per_block": round(dt * 1e6, 2), "launches_us_per_block": round(fir * 1e3, 2), "Msps_all_clients": round(n * S / dt / 1e6, 0),
        "algorithmic_GBs_launch": round(n * S * bpu / (fir * 1e-3) / 1e9, 0), "hbm_model_frac_launch": round(n * S * bpu / (fir * 1e-3) / 8e12, 3),
        "note": "per-client-read model (8 B in + 8/D B out per client and sample); the block is read once and shared, so the model number passes 1"}
print(json.dumps(out, indent=1))
