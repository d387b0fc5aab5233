#!/usr/bin/env python3
"""tools/replay_iq.py -- end-to-end run of the batched path without the TCP server: replay a raw IQ file through
wire-format admission (include/xlating_wire.h), the batch engine (include/xlating_batch.h) and the output sinks
(include/xlating_sinks.h).  What the reference does with a dongle + N connected clients, with the file standing in for
the device (its device plugins deliver blocks of `buffer_size` bytes: cu8 rtl-sdr, cs8 hackrf, cs16 airspy;
src/sdr/*_device.c, src/tcp_server.c:257-271).

  python tools/replay_iq.py capture.cu8 --format cu8 --band-rate 2016000 --band-freq 460100000 --out /tmp/out \\
         --client 460112000:48000 --client 460050000:96000 [--gzip] [--variant optimized]

Every --client CENTER_HZ:RATE_HZ goes through the 15-byte wire request, the admission rules and
xlating_wire_add_client, and ends up as <out>/<id>.cf32[.gz]."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import sdr_server_amd as xl  # noqa: E402

DTYPES = {"cu8": np.uint8, "cs8": np.int8, "cs16": np.int16}


def replay(path, fmt, band_rate, band_freq, clients, out_dir, buffer_size=262144, lpf_cutoff_rate=5, variant="optimized",
           gzip=False, writer_threads=2):
    """clients: [(center_hz, rate_hz)].  Returns {client_id: (center_hz, rate_hz)} of the admitted ones and the list of
    (center, rate, failure_details) of the rejected ones."""
    os.makedirs(out_dir, exist_ok=True)
    eng = xl.BatchEngine(band_rate, fmt, buffer_size)
    sinks = xl.Sinks(writer_threads=writer_threads, queue_bytes=64 * (buffer_size // 2 // 8 + 64) * 8)
    admitted, rejected = {}, []
    for center, rate in clients:
        code, req = xl.wire_parse_request(xl.wire_build_request(center, rate, band_freq, 0)[2:])
        assert code == 0
        code, adm, why = xl.wire_admit(req, band_rate, band_freq if admitted else 0, lpf_cutoff_rate)
        if code != 0:
            rejected.append((center, rate, why))
            continue
        cid = xl.wire_add_client(eng, adm, band_rate)
        if cid < 0:
            rejected.append((center, rate, 3))  # INTERNAL_ERROR
            continue
        assert sinks.attach_file(cid, out_dir, use_gzip=gzip) == 0
        admitted[cid] = (center, rate)
    elem = np.dtype(DTYPES[fmt]).itemsize
    per_block = buffer_size // elem  # the devices deliver buffer_size BYTES per callback
    data = np.fromfile(path, dtype=DTYPES[fmt])
    nblocks = 0
    for off in range(0, data.size, per_block):
        blk = data[off:off + per_block]
        if blk.size % 2:
            blk = blk[:-1]
        if blk.size == 0:
            break
        eng.process_host(blk, variant)
        eng.fetch()
        sinks.submit(eng)
        for cid in sinks.failed():
            sinks.detach(cid)
            eng.remove_client(cid)
            admitted.pop(cid, None)
        nblocks += 1
    sinks.flush()
    for cid in list(admitted):
        sinks.detach(cid)
    written, dropped = sinks.stats()
    sinks.close()
    eng.close()
    return admitted, rejected, {"blocks": nblocks, "bytes_written": written, "blocks_dropped": dropped}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("iq_file")
    ap.add_argument("--format", default="cu8", choices=list(DTYPES))
    ap.add_argument("--band-rate", type=int, default=2016000)
    ap.add_argument("--band-freq", type=int, required=True)
    ap.add_argument("--client", action="append", default=[], help="CENTER_HZ:RATE_HZ (repeatable)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--buffer-size", type=int, default=262144)
    ap.add_argument("--lpf-cutoff-rate", type=int, default=5)
    ap.add_argument("--variant", default="optimized", choices=["native", "optimized"])
    ap.add_argument("--gzip", action="store_true")
    a = ap.parse_args()
    clients = [tuple(int(v) for v in c.split(":")) for c in a.client]
    adm, rej, st = replay(a.iq_file, a.format, a.band_rate, a.band_freq, clients, a.out, a.buffer_size, a.lpf_cutoff_rate,
                          a.variant, a.gzip)
    for cid, (c, r) in adm.items():
        print(f"client {cid}: center {c} Hz rate {r} Hz -> {a.out}/{cid}.cf32{'.gz' if a.gzip else ''}")
    for c, r, why in rej:
        print(f"rejected: center {c} Hz rate {r} Hz (details {why})")
    print(st)


if __name__ == "__main__":
    main()
