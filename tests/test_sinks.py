"""CPU tests (-m "not gpu") of the per-client output sinks (include/xlating_sinks.h):
what reaches a file / gzip file / socket must be exactly the client's cf32 stream, as the reference's
write_to_file / write_to_socket would have delivered it (src/dsp_worker.c:10-39, 126-144), and a client whose
peer fails or does not keep up is dropped (dsp_worker.c:20-24, 84-86)."""
import gzip
import os
import socket
import threading
import time

import numpy as np

import sdr_server_amd as xl


def blocks_for(cid, nblocks, n=3121):
    rng = np.random.default_rng(1000 + cid)
    return [(rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) for _ in range(nblocks)]


def test_file_and_gzip_sinks_carry_the_exact_stream(tmp_path):
    s = xl.Sinks(writer_threads=3, queue_bytes=8 * 25600)
    want = {}
    for cid in range(7):
        assert s.attach_file(cid, tmp_path, use_gzip=(cid % 2 == 1)) == 0
        want[cid] = blocks_for(cid, 5, n=3121 - cid)
    assert s.attach_file(3, tmp_path) == -17  # EEXIST
    for k in range(5):
        for cid in range(7):
            assert s.write(cid, want[cid][k]) == 0
        s.flush()
    assert s.write(99, want[0][0]) == -2  # ENOENT
    for cid in range(7):
        assert s.detach(cid) == 0
        path = tmp_path / (f"{cid}.cf32.gz" if cid % 2 == 1 else f"{cid}.cf32")  # dsp_worker.c:126-144 naming
        raw = gzip.open(path, "rb").read() if cid % 2 == 1 else path.read_bytes()
        assert raw == b"".join(b.tobytes() for b in want[cid]), cid
    written, dropped = s.stats()
    assert written == sum(b.nbytes for bl in want.values() for b in bl) and dropped == 0
    assert s.failed() == []
    s.close()


def test_socket_sink_delivers_in_order_and_peer_close_fails_only_that_client():
    s = xl.Sinks(writer_threads=2, queue_bytes=64 * 25600)
    pairs = {cid: socket.socketpair() for cid in range(4)}
    got = {cid: bytearray() for cid in pairs}

    def reader(cid):
        r = pairs[cid][1]
        while True:
            d = r.recv(65536)
            if not d:
                break
            got[cid] += d
            if cid == 2 and len(got[cid]) > 30000:  # client 2 goes away mid-stream
                r.close()
                break

    th = [threading.Thread(target=reader, args=(cid,)) for cid in pairs]
    [t.start() for t in th]
    for cid, (w, _) in pairs.items():
        assert s.attach_fd(cid, w.fileno()) == 0
    want = {cid: blocks_for(cid, 12) for cid in pairs}
    failed = set()
    for k in range(12):
        for cid in pairs:
            rc = s.write(cid, want[cid][k])
            assert rc == 0 or (cid == 2 and rc == -32), (cid, rc)  # -EPIPE once the sink has failed
        time.sleep(0.01)
        failed |= set(s.failed())
    s.flush()
    failed |= set(s.failed())
    assert failed == {2}
    for cid in pairs:
        assert s.detach(cid) == 0
        pairs[cid][0].close()
    [t.join(10) for t in th]
    for cid in (0, 1, 3):
        assert bytes(got[cid]) == b"".join(b.tobytes() for b in want[cid]), cid
    assert s.stats()[1] > 0  # client 2's later blocks were dropped
    s.close()


def test_slow_peer_overflows_its_queue_and_is_dropped():
    """Back-pressure rule: the queue is bounded; a client that does not drain it is failed (the reference blocks the
    client's own dsp thread and ends up dropping its blocks; the batched path must never block on one peer)."""
    s = xl.Sinks(writer_threads=1, queue_bytes=4 * 25000)
    a, b = socket.socketpair()
    a.setsockopt(socket.SOL_SOCKET, socket.SO_SNDBUF, 4096)
    assert s.attach_fd(5, a.fileno()) == 0
    blk = blocks_for(5, 1)[0]
    codes = [s.write(5, blk) for _ in range(200)]  # nobody reads from b
    assert codes[0] == 0 and -32 in codes
    assert s.failed() == [5] and s.failed() == []  # reported once
    assert s.detach(5) == 0
    a.close()
    b.close()
    s.close()


def test_fast_writer_pool_many_clients(tmp_path):
    """1024 clients x 8 blocks of 3121 samples through 4 writer threads into raw files: byte totals and one spot-checked
    file; prints the delivery rate (the real-time need of 1024 x 48 kHz cf32 clients is 393 MB/s)."""
    n_clients, nblk = 1024, 8
    s = xl.Sinks(writer_threads=4, queue_bytes=16 * 25600)
    for cid in range(n_clients):
        assert s.attach_file(cid, tmp_path) == 0
    blk = blocks_for(7, nblk)
    t0 = time.perf_counter()
    for k in range(nblk):
        for cid in range(n_clients):
            assert s.write(cid, blk[k]) == 0
        s.flush()
    dt = time.perf_counter() - t0
    total = n_clients * sum(b.nbytes for b in blk)
    assert s.stats() == (total, 0)
    for cid in range(n_clients):
        s.detach(cid)
    assert (tmp_path / "513.cf32").read_bytes() == b"".join(b.tobytes() for b in blk)
    print(f"sinks: {total / dt / 1e6:.0f} MB/s into {n_clients} files with 4 writer threads")
    s.close()


def test_stalled_peer_does_not_starve_the_other_clients_of_its_writer_thread(tmp_path):
    """ONE writer thread serves a socket whose peer never reads and three healthy clients (a socket that is read and two
    files).  The stalled client parks its bytes; the healthy ones keep receiving every block in time -- only the stalled
    client is ever failed, and only by its own queue overflow (advice r1: a blocking write to the stalled peer used to
    hold the thread until the healthy clients' queues overflowed too)."""
    s = xl.Sinks(writer_threads=1, queue_bytes=6 * 25000)
    stalled_w, stalled_r = socket.socketpair()
    stalled_w.setsockopt(socket.SOL_SOCKET, socket.SO_SNDBUF, 4096)
    ok_w, ok_r = socket.socketpair()
    got = bytearray()

    def reader():
        while True:
            d = ok_r.recv(65536)
            if not d:
                break
            got.extend(d)

    th = threading.Thread(target=reader)
    th.start()
    assert s.attach_fd(0, stalled_w.fileno()) == 0
    assert s.attach_fd(1, ok_w.fileno()) == 0
    assert s.attach_file(2, tmp_path) == 0 and s.attach_file(3, tmp_path, use_gzip=True) == 0
    want = {cid: blocks_for(cid, 40) for cid in (0, 1, 2, 3)}
    codes = {cid: [] for cid in want}
    for k in range(40):
        for cid in want:
            codes[cid].append(s.write(cid, want[cid][k]))
        time.sleep(0.005)  # a block period; the healthy clients' queues (6 blocks) must never fill up
    assert all(c == 0 for cid in (1, 2, 3) for c in codes[cid]), {cid: codes[cid] for cid in (1, 2, 3)}
    assert codes[0][0] == 0 and -32 in codes[0]  # the stalled client overflowed its own queue and was dropped
    assert s.failed() == [0]
    t0 = time.perf_counter()
    assert s.detach(0) == 0  # does not wait for the dead peer
    assert time.perf_counter() - t0 < 1.0
    s.flush()
    for cid in (1, 2, 3):
        assert s.detach(cid) == 0
    ok_w.close()
    th.join(10)
    assert bytes(got) == b"".join(b.tobytes() for b in want[1])
    assert (tmp_path / "2.cf32").read_bytes() == b"".join(b.tobytes() for b in want[2])
    assert gzip.open(tmp_path / "3.cf32.gz", "rb").read() == b"".join(b.tobytes() for b in want[3])
    for sk in (stalled_w, stalled_r, ok_r):
        sk.close()
    s.close()


def test_detach_cuts_off_a_peer_that_stopped_reading():
    """detach of a NON-failed sink whose peer has stopped reading returns after a bounded wait (advice r1)."""
    s = xl.Sinks(writer_threads=1, queue_bytes=64 * 25000)
    a, b = socket.socketpair()
    a.setsockopt(socket.SOL_SOCKET, socket.SO_SNDBUF, 4096)
    assert s.attach_fd(9, a.fileno()) == 0
    blk = blocks_for(9, 1)[0]
    for _ in range(8):
        assert s.write(9, blk) == 0  # fits the queue, but the peer takes only a few KB
    t0 = time.perf_counter()
    assert s.detach(9) == 0
    assert time.perf_counter() - t0 < 4.0
    a.close()
    b.close()
    s.close()
