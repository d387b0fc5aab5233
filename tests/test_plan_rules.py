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
unsigned reserve(unsigned wgs) { return xl_chain_reserve_per_xcd(wgs, wgs); }
unsigned reserve2(unsigned wgs, unsigned load_wgs) { return xl_chain_reserve_per_xcd(wgs, load_wgs); }
unsigned launch_ps(unsigned M, unsigned K, unsigned V, unsigned Dpad, unsigned mix, unsigned G) { return xl_client_launch_ps(M, K, V, Dpad, mix, G); }
unsigned chain_ns(unsigned K) { return xl_chain_block_ns(K); }
unsigned load_wgs(double ps_sum, unsigned kmax) { return xl_plan_load_wgs(ps_sum, kmax); }
int band(unsigned load_wgs) { return xl_chain_band(load_wgs); }
unsigned hyst(unsigned load_wgs, int last_band) { return xl_chain_load_with_hysteresis(load_wgs, last_band); }
"""


def _lib(tmp_path):
    src = tmp_path / "rules.c"
    src.write_text(SHIM)
    so = str(tmp_path / "rules.so")
    subprocess.run(["gcc", "-std=c11", "-O1", "-shared", "-fPIC", "-I", os.path.join(ROOT, "sdr-server_amd", "csrc"), str(src), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.load_wgs.argtypes = [ctypes.c_double, ctypes.c_uint]
    return lib


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


def test_chain_reservation_follows_the_plans_load(tmp_path):
    """The bands are measured on the server-default shape; a plan of another shape enters them with its launch time per unit of chain
    time, expressed as the default-shape client count that has the same ratio (xl_plan_load_wgs)."""
    lib = _lib(tmp_path)
    # the default shape (2.016 Msps cu8 -> 48 kHz, 505 taps: M = 128, K = 3121, V = 116, 48 padded branches, two-half mix, 8 blocks per
    # call): per client 18.7 ns of launches against 25.7 us of chain per block -- the measured 7.1 (mix) + 11.7 (inverse) ns and 24.6 us
    d = lib.launch_ps(128, 3121, 116, 48, 1, 8)
    assert 18000 <= d <= 19500 and 25000 <= lib.chain_ns(3121) <= 26500, (d, lib.chain_ns(3121))
    for clients in (64, 1000, 1024, 2048, 2049, 3008, 3009, 4096, 8192):
        assert lib.load_wgs(float(clients) * d, 3121) == -(-clients // 64), clients  # its own client count
    # BASELINE config 5 (cf32 10 Msps, D = 100, 257 taps: K = 1311, V = 126, 104 padded branches, float32 mix): 2.2 x the load per client
    c5 = lib.launch_ps(128, 1311, 126, 104, 3, 8)
    assert 17000 <= c5 <= 20500, c5  # measured: mix 14.0 + inverse 5.6 ns per (client, block)
    ratio = (c5 / lib.chain_ns(1311)) / (d / lib.chain_ns(3121))
    assert 2.0 <= ratio <= 2.5, ratio
    want = {512: 1, 1024: 0, 2048: 0, 4096: 0}  # CUs per XCD: reserved at 512 clients (measured better), none beyond (measured better)
    for clients, per_xcd in want.items():
        wgs, load = -(-clients // 64), lib.load_wgs(float(clients) * c5, 1311)
        assert lib.reserve2(wgs, load) == per_xcd, (clients, load)
    # rounds are for plans whose clients weigh about what the measured shape's do
    assert lib.reserve2(64, 64) == 4 and lib.reserve2(64, 95) == 3 and lib.reserve2(64, 97) == 0
    # a plan whose launches are short against the recurrence (few taps, small decimation) keeps one CU per chain workgroup
    light = lib.launch_ps(128, 3121, 126, 8, 1, 8)
    assert lib.reserve2(16, lib.load_wgs(1024.0 * light, 3121)) == 2


def test_reservation_bands_have_hysteresis(tmp_path):
    """The rule is not monotonic (CUs up to 32 chain workgroups, none for 33..47, rounds from 48) and a change of the reservation
    re-creates the CU-masked stream pair (~25 ms + a full synchronisation): a plan stays in the band of the previous plan until its load
    is XL_BAND_HYST = 2 workgroups past an edge, so that a population hovering at 2048 / 2049 or 3008 / 3009 clients keeps its streams."""
    lib = _lib(tmp_path)
    assert [lib.band(n) for n in (1, 32, 33, 47, 48, 200)] == [0, 0, 1, 1, 2, 2]
    for n in range(1, 300):
        assert lib.hyst(n, -1) == n and lib.hyst(n, lib.band(n)) == n  # no history / no edge crossed: the load itself
    # hovering at an edge: the band of the previous plan is kept
    band = lib.band(32)
    seen = []
    for clients in (2048, 2049, 2048, 2112, 2049, 2176, 2048):  # 32, 33, 32, 33, 33, 34, 32 workgroups
        load = lib.hyst(-(-clients // 64), band)
        band = lib.band(load)
        seen.append((band, lib.reserve2(-(-clients // 64), load)))
    assert [b for b, _ in seen] == [0] * 7 and all(r >= 4 for _, r in seen), seen
    # two workgroups past the edge: the band changes, and the way back has the same margin
    assert lib.band(lib.hyst(35, 0)) == 1 and lib.band(lib.hyst(34, 0)) == 0
    assert lib.band(lib.hyst(31, 1)) == 1 and lib.band(lib.hyst(30, 1)) == 0
    assert lib.band(lib.hyst(48, 1)) == 1 and lib.band(lib.hyst(49, 1)) == 1 and lib.band(lib.hyst(50, 1)) == 2
    assert lib.band(lib.hyst(47, 2)) == 2 and lib.band(lib.hyst(46, 2)) == 2 and lib.band(lib.hyst(45, 2)) == 1
