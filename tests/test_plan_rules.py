"""CPU test (-m "not gpu") of the plan's two measured size rules (sdr-server_amd/csrc/xl_plan_rules.h, plain C): which inverse kernel a
polyphase launch takes by its tile count, and how many CUs per XCD the side-stream recurrence kernel is given by its workgroup count.
The GPU suite checks that the engine APPLIES them (describe() at 1024 / 4096 clients); this pins the numbers themselves."""
import ctypes
import os
import subprocess

from conftest import ROOT

SHIM = r"""
#include "xl_plan_rules.h"
unsigned pick(unsigned M, unsigned opt, unsigned tiles) { return xlp_inverse_pick(M, opt, tiles); }
unsigned rounds(unsigned wgs) { return xl_chain_rounds(wgs); }
unsigned reserve(unsigned wgs) { return xl_chain_reserve_per_xcd(wgs); }
"""


def _lib(tmp_path):
    src = tmp_path / "rules.c"
    src.write_text(SHIM)
    so = str(tmp_path / "rules.so")
    subprocess.run(["gcc", "-std=c11", "-O1", "-shared", "-fPIC", "-I", os.path.join(ROOT, "sdr-server_amd", "csrc"), str(src), "-o", so], check=True)
    return ctypes.CDLL(so)


def test_inverse_kernel_by_launch_size(tmp_path):
    lib = _lib(tmp_path)
    tiles = lambda clients, blocks: -(-(3121 * blocks + 2) // 116) * (clients // 128) * 4  # server default: D = 42, V = 116
    assert tiles(1024, 1) == 864 and tiles(1024, 8) == 6912 and tiles(4096, 8) == 27648
    for clients, blocks, want in ((1024, 1, 5), (2048, 1, 5), (4096, 1, 3), (1024, 8, 3), (2048, 8, 6), (4096, 8, 6), (128, 8, 5)):
        assert lib.pick(128, 0, tiles(clients, blocks)) == want, (clients, blocks)
    assert lib.pick(128, 0, 2048) == 5 and lib.pick(128, 0, 2049) == 3 and lib.pick(128, 0, 8192) == 3 and lib.pick(128, 0, 8193) == 6
    assert lib.pick(128, 0, 2688) == 3  # BASELINE config 5 at 1024 clients x 8 blocks
    for forced in (3, 5, 6):
        for t in (1, 5000, 100000):
            assert lib.pick(128, forced, t) == forced
            assert lib.pick(256, forced, t) == 3  # 256-point classes: the LDS transform only
    assert lib.pick(256, 0, 100000) == 3


def test_chain_kernel_cu_reservation(tmp_path):
    lib = _lib(tmp_path)
    wgs = lambda clients: -(-clients // 64)
    # one CU per chain workgroup (8 XCDs) up to 32 workgroups (2048 clients); none from there to 47 workgroups (3008 clients)
    for clients, per_xcd in ((64, 1), (512, 1), (513, 2), (1024, 2), (1536, 3), (2048, 4), (2049, 0), (2560, 0), (3008, 0)):
        assert lib.rounds(wgs(clients)) == 1 and lib.reserve(wgs(clients)) == per_xcd, clients
    # in rounds beyond: 2 from 48 workgroups (3009 .. 3072 clients), 3 from 80 (5120), 4 from 112 (7168)
    for clients, rounds, per_xcd in ((3009, 2, 3), (3072, 2, 3), (4096, 2, 4), (5056, 2, 5), (5120, 3, 4), (7168, 4, 4), (8192, 4, 4)):
        assert lib.rounds(wgs(clients)) == rounds and lib.reserve(wgs(clients)) == per_xcd, clients
    # the reservation never needs more rounds than the rule allows, and never more than half the chip
    for n in range(1, 1025):
        r, c = lib.rounds(n), lib.reserve(n)
        assert c <= 16 and (c == 0 or 8 * c * r >= n) and (c > 0 or 32 < n < 48 or n > 512), n
