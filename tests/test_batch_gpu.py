"""GPU parity tests (-m gpu) of the batched fan-out engine (include/xlating_batch.h): every client's stream must
equal what a single reference filter created at the client's join time would have produced.
native: bit-exact vs the oracle;  optimized: max|d|/max|y| <= 1e-5 (BASELINE.json north_star)."""
import os

import numpy as np
import pytest

import scenarios
import siggen
import sdr_server_amd as xl
from conftest import bits_equal
from pyoracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
REL_TOL = 1e-5
FS = 2016000


def lpf(fs, cutoff, tw):
    code, t = xl.create_low_pass_filter(1.0, fs, cutoff, tw)
    assert code == 0
    return t


def rel_err(a, b):
    if len(b) == 0:
        return 0.0
    return float(np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max() / max(np.abs(b).max(), 1e-30))


def check_clients(eng, oracles, fmt, x, variant, ids=None):
    eng.process_host(x, variant)
    eng.fetch()
    for cid, o in oracles.items():
        if ids is not None and cid not in ids:
            o.process(fmt, x)  # keep the oracle's stream state in step
            continue
        want = o.process(fmt, x)
        got = eng.output(cid)
        assert eng.output_len(cid) == len(want), (cid, eng.output_len(cid), len(want))
        if variant == "native":
            assert bits_equal(got, want), f"client {cid}"
        else:
            assert rel_err(got, want) <= REL_TOL, (cid, rel_err(got, want))


@pytest.mark.parametrize("variant", ["native", "optimized"])
@pytest.mark.parametrize("nclients", [1, 13, 24])
def test_homogeneous_clients_server_default(variant, nclients):
    """N clients, 48 kHz off 2.016 Msps (D=42, 505 taps), three blocks incl. a ragged one."""
    taps = lpf(FS, 24000, 9600)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    oracles = {}
    for c in range(nclients):
        fc = -900000 + c * 28000
        cid = eng.add_client(42, taps, fc)
        oracles[cid] = Oracle(42, taps, fc, FS, 262144)
    assert eng.num_clients == nclients
    for k, n in enumerate((262144, 100002, 262144)):
        check_clients(eng, oracles, "cu8", siggen.xs_u8(siggen.XS_SEED + k, n), variant)
    eng.close()


def test_mixed_rates_config3_64_clients():
    """BASELINE config 3: 32 x 48 kHz (D=42, T=505) + 32 x 96 kHz (D=21, T=253) sharing each block."""
    t48, t96 = lpf(FS, 24000, 9600), lpf(FS, 48000, 19200)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    oracles = {}
    for c in range(64):
        fc = -900000 + c * 28000
        D, taps = (42, t48) if c % 2 == 0 else (21, t96)
        cid = eng.add_client(D, taps, fc)
        oracles[cid] = Oracle(D, taps, fc, FS, 262144)
    for k in range(2):
        check_clients(eng, oracles, "cu8", siggen.xs_u8(siggen.XS_SEED + 7 + k, 262144), "native")
    check_clients(eng, oracles, "cu8", siggen.xs_u8(siggen.XS_SEED + 9, 262144), "optimized")
    eng.close()


def test_join_and_leave_mid_stream():
    """A client added after some blocks starts a fresh stream (zero history, phase 1, own output grid); removing a
    client does not disturb the others (dsp_worker_start / dsp_worker_destroy semantics)."""
    taps = lpf(FS, 24000, 9600)
    t101 = lpf(FS, 24000, 48000)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    oracles = {}
    for c in range(9):
        cid = eng.add_client(42, taps, 1000 * c - 4000)
        oracles[cid] = Oracle(42, taps, 1000 * c - 4000, FS, 262144)
    blocks = [siggen.xs_u8(100 + k, n) for k, n in enumerate((262144, 262144, 50000, 262144, 262144))]
    check_clients(eng, oracles, "cu8", blocks[0], "native")
    # two late joiners: same shape as the others, and a different (D, T)
    cid = eng.add_client(42, taps, 777)
    oracles[cid] = Oracle(42, taps, 777, FS, 262144)
    cid = eng.add_client(42, t101, -5555)
    oracles[cid] = Oracle(42, t101, -5555, FS, 262144)
    check_clients(eng, oracles, "cu8", blocks[1], "native")
    gone = sorted(oracles)[3]
    eng.remove_client(gone)
    oracles.pop(gone).close()
    check_clients(eng, oracles, "cu8", blocks[2], "native")
    cid = eng.add_client(42, taps, 31337)  # reuses the freed slot; must start from phase 1
    oracles[cid] = Oracle(42, taps, 31337, FS, 262144)
    check_clients(eng, oracles, "cu8", blocks[3], "native")
    check_clients(eng, oracles, "cu8", blocks[4], "optimized")
    eng.close()


@pytest.mark.parametrize("fmt", ["cs8", "cs16", "cf32"])
def test_other_input_formats(fmt):
    taps = lpf(FS, 24000, 48000)  # 101 taps
    nsamp = 65536
    eng = xl.BatchEngine(FS, fmt, 4 * nsamp)
    oracles = {}
    for c in range(10):
        cid = eng.add_client(42, taps, -300000 + 61111 * c)
        oracles[cid] = Oracle(42, taps, -300000 + 61111 * c, FS, 4 * nsamp)
    for k in range(2):
        if fmt == "cs8":
            x = siggen.xs_s8(40 + k, 2 * nsamp)
        elif fmt == "cs16":
            x = siggen.xs_s16(40 + k, 2 * nsamp)
        else:
            x = (siggen.xs_s16(40 + k, 2 * nsamp).astype(np.float32) / np.float32(32768)).astype(np.float32)
        check_clients(eng, oracles, fmt, x, "native")
    eng.close()


def test_config5_cf32_10msps_257_taps():
    """BASELINE config 5 (as corrected in SURVEY D4): cf32 input at 10 Msps, D=100, 257 explicit taps."""
    taps = siggen.hamming_sinc(257, 0.004)
    nsamp = 131072
    eng = xl.BatchEngine(10000000, "cf32", 2 * nsamp)
    oracles = {}
    for c in range(16):
        fc = -4000000 + 500000 * c
        cid = eng.add_client(100, taps, fc)
        oracles[cid] = Oracle(100, taps, fc, 10000000, 2 * nsamp)
    x = np.empty(2 * nsamp, np.float32)
    x[:] = siggen.sin_f32(0, 2 * nsamp)
    check_clients(eng, oracles, "cf32", x, "native")
    check_clients(eng, oracles, "cf32", x, "optimized")
    eng.close()


def test_device_pointer_path_matches_host_path():
    """xlating_batch_process_device: block already in HBM (what an RCCL broadcast leaves), caller's stream."""
    import torch

    taps = lpf(FS, 24000, 9600)
    e1 = xl.BatchEngine(FS, "cu8", 262144)
    e2 = xl.BatchEngine(FS, "cu8", 262144)
    ids = []
    for c in range(16):
        ids.append((e1.add_client(42, taps, 5000 * c), e2.add_client(42, taps, 5000 * c)))
    for k in range(3):
        x = siggen.xs_u8(900 + k, 262144)
        e1.process_host(x, "optimized")
        xd = torch.from_numpy(x).cuda()
        e2.process_device(xd.data_ptr(), x.size, "optimized", torch.cuda.current_stream().cuda_stream)
        e1.fetch()
        e2.fetch()
        for a, b in ids:
            assert bits_equal(e1.output(a), e2.output(b))
    e1.close()
    e2.close()


@pytest.mark.parametrize("variant", ["native", "optimized"])
def test_lookahead_nco_with_ragged_blocks_and_late_fetch(variant):
    """The phase table of block k+1 is tabulated ahead assuming block k's length.  Ragged blocks (wrong guess), a
    tiny block without output, a client joining mid-stream and fetching only some blocks must all leave every
    client's stream equal to the oracle's.  Inputs arrive as device buffers on the caller's (torch) stream."""
    import torch

    taps = lpf(FS, 24000, 9600)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    oracles = {}
    for c in range(19):
        cid = eng.add_client(42, taps, -700000 + 70001 * c)
        oracles[cid] = Oracle(42, taps, -700000 + 70001 * c, FS, 262144)
    lens = [262144, 262144, 100002, 262144, 262144, 8, 262144, 262144]
    xs = [siggen.xs_u8(3000 + k, n) for k, n in enumerate(lens)]
    recv = [torch.empty(262144, dtype=torch.uint8, device="cuda") for _ in range(2)]
    stream = torch.cuda.current_stream().cuda_stream
    for k, x in enumerate(xs):
        if k == 4:
            cid = eng.add_client(42, taps, 4242)
            oracles[cid] = Oracle(42, taps, 4242, FS, 262144)
        buf = recv[k % 2]
        buf[: x.size].copy_(torch.from_numpy(x), non_blocking=False)  # reuse of the buffer is stream-ordered
        eng.process_device(buf.data_ptr(), x.size, variant, stream)
        want = {cid: o.process("cu8", x) for cid, o in oracles.items()}
        if k in (1, 2, 5, 7):
            eng.fetch()
            for cid in oracles:
                got = eng.output(cid)
                if variant == "native":
                    assert bits_equal(got, want[cid]), (k, cid)
                else:
                    assert len(got) == len(want[cid]) and rel_err(got, want[cid]) <= REL_TOL, (k, cid)
    eng.close()


@pytest.mark.parametrize("nbytes", [200, 2000, 30])
def test_small_blocks_streaming(nbytes):
    """Many tiny blocks (launches of one or two workgroups, blocks shorter than the filter): history roll, phase
    carry and the global output grid must survive (reference unit-test shape: 57 taps, D = 5, test_xlating.c:15-22)."""
    taps = lpf(48000, 4800, 2000)
    eng = xl.BatchEngine(48000, "cu8", 4000)
    oracles = {}
    for c in range(5):
        cid = eng.add_client(5, taps, -12000 + 3000 * c)
        oracles[cid] = Oracle(5, taps, -12000 + 3000 * c, 48000, 4000)
    for k in range(12):
        n = nbytes if k % 5 != 3 else nbytes + 2
        check_clients(eng, oracles, "cu8", siggen.ramp_u8(k * 977, n), "native" if k % 2 == 0 else "optimized")
    eng.close()


def test_full_size_1024_clients_properties():
    """BASELINE target size (1024 concurrent 48 kHz clients, 505 taps, 262144-byte blocks).  The oracle is too slow
    for all of it, so: (a) duplicated clients placed in different tiles/groups/XCDs must agree bit for bit,
    (b) a checksum over all outputs is identical across two engines fed the same stream, (c) 16 sampled clients
    are compared with the oracle."""
    taps = lpf(FS, 24000, 9600)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    fcs = [-984000 + 1920 * (c % 512) for c in range(1024)]
    for fc in fcs:
        eng.add_client(42, taps, fc)
    rng = np.random.default_rng(3)
    sample = sorted(rng.choice(1024, 16, replace=False).tolist())
    oracles = {c: Oracle(42, taps, fcs[c], FS, 262144) for c in sample}
    for k in range(2):
        x = siggen.xs_u8(5000 + k, 262144)
        eng.process_host(x, "native")
        eng.fetch()
        outs = [eng.output(c) for c in range(1024)]
        for c in range(512):
            assert bits_equal(outs[c], outs[c + 512]), c
        for c in sample:
            assert bits_equal(outs[c], oracles[c].process("cu8", x)), c
        assert all(len(o) == len(outs[0]) for o in outs)
    eng.close()


@pytest.mark.parametrize("nclients", [13, 24, 70])
def test_nco_riders_forced_small(nclients, monkeypatch):
    """The NCO role riding in the spare waves of partially filled workgroups (xl_kernels.hip "riders"; normally only
    chosen for ~400-1200 clients) forced on for small engines, so that the oracle can check EVERY client: the tables
    the riders tabulate ahead must be the reference's recurrence (xlating.c:70-73) bit for bit, through ragged
    block lengths (wrong length guess -> stand-alone tabulation) and a mixed-rate class."""
    monkeypatch.setenv("XL_EXP_RIDERS_MIN", "1")
    t48, t96 = lpf(FS, 24000, 9600), lpf(FS, 48000, 19200)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    oracles = {}
    for c in range(nclients):
        fc = -900000 + c * 25000
        D, taps = (21, t96) if (nclients == 70 and c % 7 == 3) else (42, t48)
        cid = eng.add_client(D, taps, fc)
        oracles[cid] = Oracle(D, taps, fc, FS, 262144)
    for k, n in enumerate((262144, 262144, 100002, 100002, 262144, 262144)):
        check_clients(eng, oracles, "cu8", siggen.xs_u8(siggen.XS_SEED + 40 + k, n), "native" if k != 4 else "optimized")
    eng.close()


def test_1000_clients_split_group_riders():
    """1000 clients = exactly 25 full groups of 4 x 10: the planner splits the last group 3 + 1 to make room for the
    riders.  Duplicates must agree bit for bit and 12 sampled clients must match the oracle over three blocks."""
    taps = lpf(FS, 24000, 9600)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    fcs = [-984000 + 1920 * (c % 500) for c in range(1000)]
    for fc in fcs:
        eng.add_client(42, taps, fc)
    sample = [0, 9, 10, 499, 500, 959, 960, 969, 970, 989, 990, 999]
    oracles = {c: Oracle(42, taps, fcs[c], FS, 262144) for c in sample}
    for k in range(3):
        x = siggen.xs_u8(7000 + k, 262144)
        eng.process_host(x, "native")
        eng.fetch()
        outs = [eng.output(c) for c in range(1000)]
        for c in range(500):
            assert bits_equal(outs[c], outs[c + 500]), c
        for c in sample:
            assert bits_equal(outs[c], oracles[c].process("cu8", x)), c
    eng.close()


# ---------------------------------------------------------------------------------------------------------------
# Polyphase overlap-save path (xl_polyphase.hip): the optimized arithmetic of big classes.  Same operator as the
# direct kernels, so the same oracle and the same 1e-5 bar; XL_EXP_POLY=1 forces it for classes of any size so
# that the oracle can check every client.
# The transform length M is 128 for filters of up to 32 taps per branch and 256 beyond; XL_EXP_POLY_M forces either, and
# the forced-path tests run with both.
@pytest.fixture(params=[(128, 0, 1), (128, 3, 1), (256, 0, 1), (128, 5, 3), (256, 0, 3), (128, 3, 3), (128, 6, 1), (128, 6, 3), (64, 0, 1), (64, 0, 3)],
                ids=["M128", "M128-lds-inverse", "M256", "M128-f32-mix", "M256-f32-mix", "M128-lds-inverse-f32-mix", "M128-cut32-inverse",
                     "M128-cut32-inverse-f32-mix", "M64", "M64-f32-mix"])
def poly_m(request, monkeypatch):
    """Transform length of the forced polyphase plan; at M = 128 the inverse launch's transform in the registers of eight lanes per
    column (option "inverse_kernel" = 5: xlp_inverse8_kernel -- what the default, 0, picks for launches as small as these), cut 32 x 4
    (6: xlp_inverse32_kernel, the default's pick for big launches) or staged in LDS on swizzled rows (3: xlp_inverse_kernel);
    the mix launch on the matrix cores with two-half float16 operands where the class allows them and float32 operands elsewhere
    (option "mix_kernel" = 1, the default) or with float32 operands for every class (3)."""
    m, inv, mix = request.param
    monkeypatch.setenv("XL_EXP_POLY_M", str(m))
    monkeypatch.setenv("XL_EXP_INV", str(inv))
    monkeypatch.setenv("XL_EXP_MIX", str(mix))
    return m


def _poly_engine(monkeypatch, fmt="cu8", max_input=262144, fs=FS):
    monkeypatch.setenv("XL_EXP_POLY", "1")
    return xl.BatchEngine(fs, fmt, max_input)


def test_polyphase_forced_server_default_ragged_and_join(monkeypatch, poly_m):
    """D=42, 505 taps: ragged block lengths (output grid offset changes every block), a client joining mid-stream
    (zero history below its join point, phase 1), a native block in between (both paths share history and phases)."""
    taps = lpf(FS, 24000, 9600)
    eng = _poly_engine(monkeypatch)
    oracles = {}
    for c in range(20):
        fc = -900000 + c * 91000
        oracles[eng.add_client(42, taps, fc)] = Oracle(42, taps, fc, FS, 262144)
    assert "polyphase: cls0 D42 T505 cols20 V%d M%d" % (poly_m - 12, poly_m) in eng.describe(), eng.describe()
    worst = 0.0
    # (blocks of 2000 and 30 bytes hold fewer than 512 outputs: those fall back to the direct kernel in between)
    for k, n in enumerate((262144, 262144, 100002, 100002, 262144, 50000, 262144, 2000, 262144, 30, 262144)):
        if k == 3:
            oracles[eng.add_client(42, taps, 123456)] = Oracle(42, taps, 123456, FS, 262144)
        if k == 5:  # a client leaves: the plan (columns, branch spectra) is rebuilt, everybody else streams on
            gone = sorted(oracles)[7]
            eng.remove_client(gone)
            oracles.pop(gone).close()
        x = siggen.xs_u8(siggen.XS_SEED + 80 + k, n)
        variant = "native" if k == 4 else "optimized"
        eng.process_host(x, variant)
        eng.fetch()
        for cid, o in oracles.items():
            want = o.process("cu8", x)
            got = eng.output(cid)
            assert len(got) == len(want)
            if variant == "native":
                assert bits_equal(got, want)
            else:
                worst = max(worst, rel_err(got, want))
    assert worst <= REL_TOL, worst
    eng.close()


def test_polyphase_forced_mixed_rates_and_fixture_shape(monkeypatch, poly_m):
    """Two classes on the path at once (48 kHz: D=42/T=505, 96 kHz: D=21/T=253), then the reference's own test
    shape (test_xlating.c: 57 taps, D=5, fs 48000, fc -12000) on a long ramp input."""
    t48, t96 = lpf(FS, 24000, 9600), lpf(FS, 48000, 19200)
    eng = _poly_engine(monkeypatch)
    oracles = {}
    for c in range(40):
        fc = -700000 + c * 35000
        D, taps = (42, t48) if c % 2 == 0 else (21, t96)
        oracles[eng.add_client(D, taps, fc)] = Oracle(D, taps, fc, FS, 262144)
    for k in range(3):
        check_clients(eng, oracles, "cu8", siggen.xs_u8(siggen.XS_SEED + 90 + k, 262144 if k != 1 else 131074), "optimized")
    eng.close()
    code, t57 = xl.create_low_pass_filter(1.0, 48000, 4800, 2000)
    assert code == 0 and len(t57) == 57
    eng = _poly_engine(monkeypatch, max_input=100000, fs=48000)
    oracles = {}
    for fc in (-12000, 0, 7000):
        oracles[eng.add_client(5, t57, fc)] = Oracle(5, t57, fc, 48000, 100000)
    assert "polyphase: cls0 D5 T57 cols3 V%d M%d" % (poly_m - 11, poly_m) in eng.describe(), eng.describe()
    for k in range(3):
        check_clients(eng, oracles, "cu8", siggen.ramp_u8(k * 31, 100000 - 2 * k), "optimized")
    eng.close()


@pytest.mark.parametrize("fmt", ["cs8", "cs16", "cf32"])
def test_polyphase_forced_other_formats(fmt, monkeypatch, poly_m):
    taps = lpf(FS, 24000, 9600)
    n = 65536
    eng = _poly_engine(monkeypatch, fmt=fmt)
    oracles = {}
    for c in range(6):
        fc = -500000 + c * 200000
        oracles[eng.add_client(42, taps, fc)] = Oracle(42, taps, fc, FS, 262144)
    for k in range(2):
        if fmt == "cs8":
            x = siggen.xs_s8(4000 + k, 2 * n)
        elif fmt == "cs16":
            x = siggen.xs_s16(4000 + k, 2 * n)
        else:
            x = (siggen.xs_s16(4000 + k, 2 * n).astype(np.float32) / 32768.0).astype(np.float32)
        check_clients(eng, oracles, fmt, x, "optimized")
    eng.close()


@pytest.mark.parametrize("shape", ["perf_2429_taps", "cf32_d100_257_taps", "cs16_d100_257_taps", "cu8_d112_589_taps", "d400_4819_taps"])
def test_polyphase_forced_other_shapes(shape, monkeypatch, poly_m):
    """Other branch counts / taps per branch on the path: the reference's perf shape (test/perf_xlating.c: 2429 taps,
    D=42 -> 58 taps per branch, 199 valid outputs per segment), BASELINE config 5 (cf32 in, D=100, 257 taps -> 3 taps
    per branch: 13 k-blocks on the two-half mix, segment scales), the same shape off an Airspy-style cs16 stream (src/xlating.c:374-382)
    and D = 112 (14 k-blocks: the most the two-half kernel holds), and a huge decimation (20 Msps -> 50 kHz: D=400, 4819 taps: float32
    operands streamed per pass)."""
    if shape == "perf_2429_taps":
        fs, D, fmt, nbytes = FS, 42, "cu8", 262144
        taps = lpf(fs, 24000, 2000)
        assert len(taps) == 2429
        x = [siggen.staircase_u8(nbytes), siggen.xs_u8(77, nbytes - 4)]
    elif shape == "cf32_d100_257_taps":
        fs, D, fmt, nbytes = 10000000, 100, "cf32", 262144
        taps = siggen.hamming_sinc(257, 0.004)
        x = [(siggen.xs_s16(300 + k, 131072).astype(np.float32) / np.float32(32768)).astype(np.float32) for k in range(2)]
    elif shape == "cs16_d100_257_taps":
        fs, D, fmt, nbytes = 10000000, 100, "cs16", 262144
        taps = siggen.hamming_sinc(257, 0.004)
        x = [siggen.xs_s16(310 + k, 131072) for k in range(2)]
    elif shape == "cu8_d112_589_taps":
        fs, D, fmt, nbytes = 5376000, 112, "cu8", 262144
        taps = lpf(fs, 24000, 22000)
        assert len(taps) == 589
        x = [siggen.xs_u8(320 + k, nbytes) for k in range(2)]
    else:
        fs, D, fmt, nbytes = 20000000, 400, "cu8", 1048576
        taps = lpf(fs, 25000, 10000)
        assert len(taps) == 4819
        x = [siggen.xs_u8(500 + k, nbytes) for k in range(2)]
    eng = _poly_engine(monkeypatch, fmt=fmt, max_input=nbytes, fs=fs)
    oracles = {}
    for c in range(5):
        fc = int(-0.3 * fs + c * 0.15 * fs)
        oracles[eng.add_client(D, taps, fc)] = Oracle(D, taps, fc, fs, nbytes)
    assert "polyphase: cls0 D%d T%d cols5" % (D, len(taps)) in eng.describe(), eng.describe()
    want_mix = "mix=mf32" if (D > 112 or os.environ.get("XL_EXP_MIX") == "3") else "mix=mfma"
    assert want_mix in eng.describe(), eng.describe()
    for xb in x:
        check_clients(eng, oracles, fmt, xb, "optimized")
    eng.close()


@pytest.mark.parametrize("m,mix", [(128, 1), (256, 1), (128, 3)], ids=["M128", "M256", "M128-f32-mix"])
def test_polyphase_matrix_core_mix_tap_scales_and_full_scale_input(m, mix, monkeypatch):
    """The matrix-core mix carries every operand as two halves after a power-of-two scale (per column for the branch
    spectra, fixed for the shared spectra): one class whose members' taps differ by 10^8 in gain (column scales 2^-2 ..
    2^25), a one-tap-dominated and an asymmetric filter among them, on full-scale inputs (constant +127:
    the largest possible spectrum value, 128 x sqrt 2 x M / 128; a full-scale square wave; noise) -- each client within the
    same 1e-5 of ITS output scale as on the FP32 path."""
    monkeypatch.setenv("XL_EXP_POLY_M", str(m))
    monkeypatch.setenv("XL_EXP_MIX", str(mix))
    base = np.asarray(lpf(FS, 24000, 9600), dtype=np.float32)
    T = len(base)
    spike = base.copy()
    spike[T // 2] += np.float32(40.0)  # one huge tap: the bound of its branch dominates the column scale
    asym = (base * np.linspace(0.2, 1.8, T).astype(np.float32)).astype(np.float32)
    variants = [base * np.float32(g) for g in (1e-4, 1.0, 37.5, 3000.0, 1e4)] + [spike, asym]
    eng = _poly_engine(monkeypatch)
    oracles = {}
    for c, taps in enumerate(variants):
        for fc in (-600000 + 170000 * c, 250000 - 31000 * c):
            t = [float(v) for v in taps]
            oracles[eng.add_client(42, t, fc)] = Oracle(42, t, fc, FS, 262144)
    assert ("mix=mf32" if mix == 3 else "mix=mfma") in eng.describe(), eng.describe()
    n = 262144
    full = np.full(n, 255, dtype=np.uint8)
    square = np.where((np.arange(n) // 2) % 84 < 42, 255, 0).astype(np.uint8)
    for x in (full, square, siggen.xs_u8(5100, n), full):
        check_clients(eng, oracles, "cu8", x, "optimized")
    eng.close()


@pytest.mark.parametrize("mix", [1, 3], ids=["mfma", "f32-mfma"])
@pytest.mark.parametrize("D,fs", [(12, 576000), (33, 1584000), (50, 2400000), (64, 3072000)])
def test_polyphase_matrix_core_mix_other_branch_counts(D, fs, mix, monkeypatch):
    """The two-half matrix-core mix is built per number of k-blocks of 8 branches (1..8): the server default is 6 (D = 42), the
    fixture shapes cover 1 (D = 5) and 3 (D = 21); here 2, 5, 7 and 8 (the last two keep a few operand registers in scratch) -- 48 kHz
    clients off other sample rates, 12 taps per branch, both transform lengths by the size rule's forcing, every client vs the
    oracle.  The float32 matrix-core mix (mix = 3, xl_mixf32.hip) is built per number of k-blocks whose operands stay in registers
    (1..14; D = 100 and D = 400 -- the streaming kernel -- run in test_polyphase_forced_other_shapes)."""
    monkeypatch.setenv("XL_EXP_MIX", str(mix))
    taps = lpf(fs, 24000, fs // 210)
    assert len(taps) >= 9 * D // 2
    n = 131072
    for m in (128, 256):
        monkeypatch.setenv("XL_EXP_POLY_M", str(m))
        eng = _poly_engine(monkeypatch, max_input=2 * n, fs=fs)
        oracles = {}
        for c in range(37):
            fc = int(-0.4 * fs + c * 0.021 * fs)
            oracles[eng.add_client(D, taps, fc)] = Oracle(D, taps, fc, fs, 2 * n)
        assert ("mix=mf32" if mix == 3 else "mix=mfma") in eng.describe() and " M%d " % m in eng.describe(), eng.describe()
        for k in range(3):
            check_clients(eng, oracles, "cu8", siggen.xs_u8(5300 + k, 2 * n if k != 1 else 2 * n - 1234), "optimized")
        eng.close()
        for o in oracles.values():
            o.close()


def test_size_rule_of_matrix_core_classes(monkeypatch):
    """The engine's own plan (no forcing): classes whose mix launch runs on the matrix cores take the polyphase path from 32
    clients and 2 taps per branch on (101 taps at D = 42: 3 per branch) -- cf32 streams too, since their mix launch multiplies
    float32 operands on the matrix cores (round 5; before: 128 clients, 4.5 taps per branch) --, the tuning knob XL_EXP_POLY_MIN
    moves the client threshold -- and the 101-tap class and the cf32 class match the oracle."""
    t101 = lpf(FS, 24000, 48000)
    assert len(t101) == 101
    eng = xl.BatchEngine(FS, "cu8", 262144, group_blocks=2)
    oracles = {}
    for c in range(40):
        fc = -800000 + 41000 * c
        oracles[eng.add_client(42, t101, fc)] = Oracle(42, t101, fc, FS, 262144)
    for c in range(20):  # (a class of 20: below the rule)
        fc = -300000 + 29000 * c
        oracles[eng.add_client(21, lpf(FS, 48000, 19200), fc)] = Oracle(21, lpf(FS, 48000, 19200), fc, FS, 262144)
    for k in range(3):
        x = siggen.xs_u8(5400 + k, 2 * 262144)
        _check_group(eng, oracles, "cu8", x, 2, "optimized")
    d = eng.describe()
    assert "polyphase: cls0 D42 T101 cols40 " in d and "mix=mfma" in d and "cls1" not in d and "optimized-mode direct: h" in d, d
    eng.close()
    monkeypatch.setenv("XL_EXP_POLY_MIN", "64")  # (a tuning knob, read when an engine is created: the smallest class that takes the path)
    eng = xl.BatchEngine(FS, "cu8", 262144, group_blocks=2)
    for c in range(40):
        eng.add_client(42, t101, -800000 + 41000 * c)
    eng.process_host_group(siggen.xs_u8(5410, 2 * 262144), 2, "optimized")
    assert "polyphase: none" in eng.describe(), eng.describe()
    eng.close()
    monkeypatch.delenv("XL_EXP_POLY_MIN")
    eng = xl.BatchEngine(FS, "cf32", 8 * 65536)
    taps = lpf(FS, 24000, 9600)
    oracles = {}
    for c in range(40):
        oracles[eng.add_client(42, taps, -800000 + 41000 * c)] = Oracle(42, taps, -800000 + 41000 * c, FS, 262144)
    for k in range(2):
        check_clients(eng, oracles, "cf32", (siggen.xs_s16(5420 + k, 2 * 65536).astype(np.float32) / np.float32(32768)).astype(np.float32), "optimized")
    # (round 6: a cf32 class takes the two-half matrix-core mix too, its spectra scaled per segment)
    assert "polyphase: cls0 D42 T505 cols40 " in eng.describe() and "mix=mfma" in eng.describe(), eng.describe()
    eng.close()


def test_size_rule_of_streamed_mix_classes():
    """Classes of more than 14 k-blocks of 8 branches (D > 112: the float32 mix with operands streamed per pass) take the polyphase path
    from 128 clients with 2 taps per branch on (round 6: measured at D = 128 / 200 / 400, profiles/r06_plan_rules_other_shapes.txt;
    rounds 4-5 asked for 4.5 taps per branch) and the direct kernel below 128 clients; D = 200, 481 taps (2.4 per branch), every client of
    two blocks against the oracle on either side of the rule."""
    fs, D = 9600000, 200
    taps = lpf(fs, 24000, 48000)
    assert 2 * D <= len(taps) < 9 * D // 2, len(taps)
    for nclients, poly in ((128, True), (96, False)):
        eng = xl.BatchEngine(fs, "cu8", 262144)
        oracles = {}
        for c in range(nclients):
            fc = int(-0.45 * fs + (0.9 * fs / nclients) * c)
            oracles[eng.add_client(D, taps, fc)] = Oracle(D, taps, fc, fs, 262144)
        for k in range(3):  # (block 0: the clients are inside their zero history -- a class of its own, direct)
            check_clients(eng, oracles, "cu8", siggen.xs_u8(5460 + k, 262144), "optimized")
        d = eng.describe()
        assert (("polyphase: cls0 D200 T%d cols%d " % (len(taps), nclients)) in d and "mix=mf32" in d) if poly else "polyphase: none" in d, d
        eng.close()


def test_matrix_core_mix_role_phases_bit_exact(monkeypatch):
    """One-block calls with the recurrence INSIDE the launches (option nco_side_stream = 0): the forward and the inverse launch carry a
    slice each of the NEXT call's NCO recurrence, the mix launch none (round 4: no launch that issues matrix instructions hosts the
    role -- a first build of xlp_mix_mfma_kernel corrupted the phases of role waves riding in it).  Three engines on the
    same stream of 120 blocks -- two-half mix, float32 mix, and the side-stream chain kernel as the reference -- : the committed
    phases of all 1024 clients agree bit for bit after every call."""
    t48 = lpf(FS, 24000, 9600)
    engs = []
    for mix, side in ((1, 0), (3, 0), (1, 1)):
        monkeypatch.setenv("XL_EXP_MIX", str(mix))
        e = xl.BatchEngine(FS, "cu8", 262144)
        e.set_option("nco_side_stream", side)
        ids = [e.add_client(42, t48, -984000 + 1920 * c) for c in range(1024)]
        engs.append((e, ids))
    assert "mix=mfma" in engs[0][0].describe() and "mix=mf32" in engs[1][0].describe() and "mix=mfma" in engs[2][0].describe()
    for k in range(120):
        x = siggen.xs_u8(7000 + k, 262144)
        ph = []
        for e, ids in engs:
            e.process_host(x, "optimized")
            e.sync()
            ph.append(np.array([e.phase(i) for i in ids], dtype=np.float32))
        for other in (1, 2):
            bad = np.flatnonzero((ph[0].view(np.uint32) != ph[other].view(np.uint32)).any(axis=1))
            assert len(bad) == 0, (k, other, bad[:32])
    for e, _ in engs:
        e.close()


def test_role_phases_soak_4096_clients_2000_one_block_calls(monkeypatch):
    """VERDICT r3 item 3's soak: 4096 clients, 2000 one-block calls with the recurrence INSIDE the launches (nco_side_stream = 0), two
    engines in one process whose launches interleave on the chip -- one with the float32 matrix-core mix, one with the two-half mix;
    both host the role in their forward and inverse launches only, stepping with scalar instructions (the packed step is what other
    launches' matrix instructions corrupted: DESIGN 3.6).  All 4096 committed phases compared bit for bit every 100 calls (a
    corrupted phase never heals: the recurrence carries it on)."""
    t48 = lpf(FS, 24000, 9600)
    engs = []
    for mix in (3, 1):
        monkeypatch.setenv("XL_EXP_MIX", str(mix))
        e = xl.BatchEngine(FS, "cu8", 262144)
        e.set_option("nco_side_stream", 0)
        ids = [e.add_client(42, t48, -984000 + 480 * c) for c in range(4096)]
        engs.append((e, ids))
    assert "mix=mf32" in engs[0][0].describe() and "mix=mfma" in engs[1][0].describe()
    blocks = [siggen.xs_u8(7300 + k, 262144) for k in range(4)]
    for k in range(2000):
        for e, _ in engs:
            e.process_host(blocks[k % 4], "optimized")
        if k % 100 == 99:
            ph = []
            for e, ids in engs:
                e.sync()
                ph.append(np.array([e.phase(i) for i in ids], dtype=np.float32))
            bad = np.flatnonzero((ph[0].view(np.uint32) != ph[1].view(np.uint32)).any(axis=1))
            assert len(bad) == 0, (k, bad[:32])
    for e, _ in engs:
        e.close()


def test_polyphase_class_next_to_direct_classes():
    """The size rule at work inside one engine: 200 x 48 kHz clients (505 taps) take the polyphase path, 9 x 96 kHz
    clients (253 taps, too few for it) stay on the direct kernel, whose launch then carries the NCO role for ALL
    clients and rolls the history; optimized and native blocks alternate.  Sampled 48 kHz clients and every 96 kHz
    client are checked against the oracle."""
    t48, t96 = lpf(FS, 24000, 9600), lpf(FS, 48000, 19200)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    cfg = [(42, t48, -950000 + 9000 * c) for c in range(200)] + [(21, t96, -800000 + 170000 * c) for c in range(9)]
    ids = [eng.add_client(D, taps, fc) for D, taps, fc in cfg]
    d = eng.describe()
    assert "polyphase: cls0 D42 T505 cols200" in d and "optimized-mode direct: h" in d, d
    sample = [0, 1, 63, 64, 127, 128, 199] + list(range(200, 209))
    oracles = {ids[i]: Oracle(cfg[i][0], cfg[i][1], cfg[i][2], FS, 262144) for i in sample}
    for k, (n, variant) in enumerate(((262144, "optimized"), (262144, "optimized"), (131074, "native"), (262144, "optimized"))):
        check_clients(eng, oracles, "cu8", siggen.xs_u8(siggen.XS_SEED + 120 + k, n), variant)
    eng.close()


def test_polyphase_100_block_drift(monkeypatch, poly_m):
    """100 consecutive blocks on the polyphase path (NCO recurrence renormalised per block, tabulated one block ahead -- on the side
    stream, or cut into two slices inside the launches): the float32 phase drift must stay the reference's own -- every 10th
    block of 8 clients against the oracle, plus the committed phases at the end."""
    taps = lpf(FS, 24000, 9600)
    eng = _poly_engine(monkeypatch)
    if os.environ.get("XL_EXP_INV") in ("3", "6") or poly_m == 256:  # (these fixtures: the recurrence inside the launches, two slices per
        eng.set_option("nco_side_stream", 0)                   # block; the others: on the side stream, the default for these calls)
    oracles = {}
    for c in range(8):
        fc = -800000 + c * 213000 + 17
        oracles[eng.add_client(42, taps, fc)] = Oracle(42, taps, fc, FS, 262144)
    worst = 0.0
    for k in range(100):
        x = siggen.xs_u8(20000 + k, 262144)
        eng.process_host(x, "optimized")
        want = {cid: o.process("cu8", x) for cid, o in oracles.items()}
        if k % 10 == 9:
            eng.fetch()
            for cid in oracles:
                worst = max(worst, rel_err(eng.output(cid), want[cid]))
    assert worst <= REL_TOL, worst
    for cid, o in oracles.items():
        pr, pi = eng.phase(cid)
        assert (np.float32(pr), np.float32(pi)) == tuple(np.float32(v) for v in o.phase), cid  # the recurrence is bit-exact
    eng.close()


def test_polyphase_default_rule_1024_clients():
    """The bench shape (1024 x 48 kHz, 505 taps): the size rule selects the path by itself.  Duplicated clients in
    different columns agree bit for bit, 16 sampled clients match the oracle within 1e-5 over three blocks, and a
    native block in between is still bit-exact (shared history / phase state)."""
    taps = lpf(FS, 24000, 9600)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    fcs = [-984000 + 1920 * (c % 512) for c in range(1024)]
    for fc in fcs:
        eng.add_client(42, taps, fc)
    assert "polyphase: cls0 D42 T505 cols1024" in eng.describe() and "optimized-mode direct: none" in eng.describe()
    rng = np.random.default_rng(11)
    sample = sorted(rng.choice(1024, 16, replace=False).tolist())
    oracles = {c: Oracle(42, taps, fcs[c], FS, 262144) for c in sample}
    for k, variant in enumerate(("optimized", "optimized", "native", "optimized")):
        x = siggen.xs_u8(9000 + k, 262144)
        eng.process_host(x, variant)
        eng.fetch()
        outs = [eng.output(c) for c in range(1024)]
        for c in range(512):
            assert bits_equal(outs[c], outs[c + 512]), c
        for c in sample:
            want = oracles[c].process("cu8", x)
            if variant == "native":
                assert bits_equal(outs[c], want), c
            else:
                assert rel_err(outs[c], want) <= REL_TOL, (c, rel_err(outs[c], want))
    assert "inv=lanes8" in eng.describe(), eng.describe()  # (864 tiles per launch: the size rule takes the 8-lane inverse kernel)
    eng.close()


def test_polyphase_plan_switches_transform_length_mid_stream():
    """A class growing past 768 clients moves from the 256-point to the 128-point plan, and back when clients leave:
    new branch spectra, new segment grid, same stream -- the only state carried over is the raw history and the phases.
    Late joiners start as a class of their own (their output grid and history differ); they join at a stream position
    that is a multiple of D, so one block later their grid and history match the old clients' and the next re-plan
    merges them.  Sampled old clients and the joiners are checked against the oracle across the switches."""
    taps = lpf(FS, 24000, 9600)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    fcs = [-984000 + 2560 * c for c in range(767)]
    ids = [eng.add_client(42, taps, fc) for fc in fcs]
    assert "cols767 V244 M256" in eng.describe(), eng.describe()
    sample = [0, 5, 300, 766]
    oracles = {ids[c]: Oracle(42, taps, fcs[c], FS, 262144) for c in sample}

    def block(k, n=262144):
        x = siggen.xs_u8(9100 + k, n)
        eng.process_host(x, "optimized")
        eng.fetch()
        for cid, o in oracles.items():
            want = o.process("cu8", x)
            got = eng.output(cid)
            assert len(got) == len(want) and rel_err(got, want) <= REL_TOL, (k, cid, rel_err(got, want))

    block(0)
    block(1, 2 * 100012)  # 131072 + 100012 samples = 42 * 5502
    joiners = []
    for j in range(2):
        cid = eng.add_client(42, taps, 111000 + 7000 * j)
        oracles[cid] = Oracle(42, taps, 111000 + 7000 * j, FS, 262144)
        joiners.append(cid)
    # (with calls behind it a changed engine keeps the plan -- and the rows -- of the latest call until the next one:
    # describe() says so instead of re-planning)
    assert "re-plan pending" in eng.describe() and "cols767 V244 M256" in eng.describe(), eng.describe()
    block(2)
    assert "classes 2" in eng.describe() and "cols767 V244 M256" in eng.describe(), eng.describe()
    eng.remove_client(ids[1])  # (a re-plan: the joiners' class now equals the old one -> 768 clients -> 128-point plan)
    block(3)
    assert "classes 1" in eng.describe() and "cols768 V116 M128" in eng.describe(), eng.describe()
    block(4, 131070)
    for cid in joiners:  # 766 clients: back to the 256-point plan
        eng.remove_client(cid)
        oracles.pop(cid).close()
    block(5)
    assert "cols766 V244 M256" in eng.describe(), eng.describe()
    eng.close()


def test_bench_group_feeder_stream_plumbing():
    """bench.py's multi-GPU feed (broadcast of super-block k+1 on a side stream while super-block k is filtered, two
    receive buffers, event-ordered reuse) with a stand-in for torch.distributed whose broadcast is the identity (this
    box has one GPU): the engine must see exactly the super-blocks in order, 8 blocks per call."""
    import torch

    import bench

    class FakeDist:
        def broadcast(self, t, src=0):
            return None

    groups = [siggen.xs_u8(8100 + k, bench.GROUP * bench.BLOCK_BYTES) for k in range(3)]
    dev = [torch.from_numpy(b).cuda() for b in groups]
    feeder = bench.GroupFeeder(torch, FakeDist(), 0, 2, dev)
    taps = lpf(FS, 24000, 48000)
    eng = xl.BatchEngine(FS, "cu8", bench.BLOCK_BYTES, group_blocks=bench.GROUP)
    oracles = {}
    for c in range(12):
        cid = eng.add_client(42, taps, bench.client_center_freq(c * 37))
        oracles[cid] = Oracle(42, taps, bench.client_center_freq(c * 37), FS, bench.BLOCK_BYTES)
    stream = torch.cuda.current_stream()
    for k in range(7):
        ptr = feeder.get(k, stream)
        eng.process_device_group(ptr, bench.BLOCK_BYTES, bench.GROUP, "optimized", stream.cuda_stream)
        feeder.consumed(k, stream)
        want = {cid: np.concatenate([o.process("cu8", bl) for bl in np.split(groups[k % 3], bench.GROUP)]) for cid, o in oracles.items()}
        if k % 3 == 2:
            eng.fetch()
            for cid in oracles:
                assert rel_err(eng.output(cid), want[cid]) <= REL_TOL, (k, cid)
    torch.cuda.synchronize()
    eng.close()


def test_sinks_submit_delivers_every_clients_stream(tmp_path):
    """xlating_sinks_submit after xlating_batch_fetch: the files hold exactly each client's output stream over the
    blocks (what dsp_worker.c's write_to_file would have written), for raw and gzip sinks; a removed client is skipped."""
    import gzip

    taps = lpf(FS, 24000, 48000)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    ids = [eng.add_client(42, taps, -400000 + 90000 * c) for c in range(9)]
    sinks = xl.Sinks(writer_threads=2, queue_bytes=16 * 25600)
    for cid in ids:
        assert sinks.attach_file(cid, tmp_path, use_gzip=(cid % 3 == 0)) == 0
    want = {cid: b"" for cid in ids}
    for k in range(4):
        if k == 2:
            eng.remove_client(ids[4])
        eng.process_host(siggen.xs_u8(siggen.XS_SEED + 200 + k, 262144 if k != 1 else 99998), "native")
        eng.fetch()
        live = [cid for cid in ids if not (k >= 2 and cid == ids[4])]
        assert sinks.submit(eng) == len(live)
        for cid in live:
            want[cid] += eng.output(cid).tobytes()
    sinks.flush()
    assert sinks.failed() == []
    for cid in ids:
        assert sinks.detach(cid) == 0
        path = tmp_path / (f"{cid}.cf32.gz" if cid % 3 == 0 else f"{cid}.cf32")
        raw = gzip.open(path, "rb").read() if cid % 3 == 0 else path.read_bytes()
        assert raw == want[cid], cid
    sinks.close()
    eng.close()


@pytest.mark.parametrize("fmt", ["cu8", "cs16"])
def test_replay_iq_file_end_to_end(fmt, tmp_path):
    """Ingest -> wire admission -> engine -> sinks (tools/replay_iq.py): a raw IQ capture file replayed in device-sized
    blocks; each admitted client's <id>.cf32 must be the stream the reference would have written for that request
    (dsp_worker.c:96-104 parameters), one bad request is rejected the way tcp_server.c answers it."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("replay_iq", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "replay_iq.py"))
    replay_iq = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(replay_iq)
    band_rate, band_freq, buffer_size = 2016000, 460100000, 262144
    nbytes = 3 * buffer_size + 50000  # a short last block
    if fmt == "cu8":
        raw = siggen.xs_u8(31337, nbytes)
    else:
        raw = siggen.xs_s16(31337, nbytes // 2)
    path = tmp_path / f"capture.{fmt}"
    raw.tofile(path)
    reqs = [(460112000, 48000), (460050000, 96000), (460100000 + 2000000, 48000), (459900000, 48000)]
    adm, rej, st = replay_iq.replay(str(path), fmt, band_rate, band_freq, reqs, str(tmp_path / "out"), buffer_size, 5, "native")
    assert sorted(adm.values()) == sorted([reqs[0], reqs[1], reqs[3]]) and rej == [(reqs[2][0], reqs[2][1], 1)]
    assert st["blocks"] == 4 and st["blocks_dropped"] == 0
    per_block = buffer_size // raw.dtype.itemsize
    for cid, (center, rate) in adm.items():
        code, taps = xl.create_low_pass_filter(1.0, band_rate, rate // 2, rate // 5)
        o = Oracle(band_rate // rate, taps, center - band_freq, band_rate, buffer_size)
        want = b"".join(o.process(fmt, raw[off:off + per_block]).tobytes() for off in range(0, raw.size, per_block))
        assert (tmp_path / "out" / f"{cid}.cf32").read_bytes() == want, cid


@pytest.mark.parametrize("seed", list(range(1, 1 + int(os.environ.get("XL_TEST_FUZZ_SEEDS", "4")))))  # (more seeds: a longer fuzz run)
@pytest.mark.parametrize("force_poly", [0, 64, 128, 256])
def test_randomised_engine_vs_oracle(seed, force_poly, monkeypatch):
    """Randomised streams: mixed decimations / tap counts (even tap counts too: the reversal quirk), clients joining and
    leaving, block lengths from a few samples to the maximum, native and optimized blocks in any order, with and
    without the polyphase path forced on.  Every client, every block, against the oracle."""
    if force_poly:
        monkeypatch.setenv("XL_EXP_POLY", "1")
        monkeypatch.setenv("XL_EXP_POLY_M", str(force_poly))
    rng = np.random.default_rng(1000 + seed)
    shapes = [(42, lpf(FS, 24000, 9600)), (21, lpf(FS, 48000, 19200)), (42, lpf(FS, 24000, 48000)),
              (7, siggen.hamming_sinc(64, 0.05)), (100, siggen.hamming_sinc(301, 0.004))]
    max_input = 131072
    eng = xl.BatchEngine(FS, "cu8", max_input)
    oracles = {}

    def join():
        D, taps = shapes[int(rng.integers(len(shapes)))]
        fc = int(rng.integers(-900000, 900000))
        oracles[eng.add_client(D, taps, fc)] = Oracle(D, taps, fc, FS, max_input)

    for _ in range(int(rng.integers(3, 14))):
        join()
    worst = 0.0
    for k in range(14):
        r = rng.random()
        if r < 0.25:
            join()
        elif r < 0.4 and len(oracles) > 1:
            gone = sorted(oracles)[int(rng.integers(len(oracles)))]
            eng.remove_client(gone)
            oracles.pop(gone).close()
        n = int(rng.choice([max_input, max_input, 100000, 50002, 2 * int(rng.integers(1, 3000)), 8]))
        x = siggen.xs_u8(int(rng.integers(1 << 30)), n)
        variant = "native" if rng.random() < 0.3 else "optimized"
        eng.process_host(x, variant)
        eng.fetch()
        for cid, o in oracles.items():
            want = o.process("cu8", x)
            got = eng.output(cid)
            assert len(got) == len(want), (k, cid, len(got), len(want))
            if variant == "native":
                assert bits_equal(got, want), (k, cid)
            else:
                worst = max(worst, rel_err(got, want))
    assert worst <= REL_TOL, worst
    eng.close()


def test_bench_block_feed_over_real_rccl():
    """bench.py's multi-GPU feed on the real RCCL backend (a world-size-1 process group on this one-GPU box, the feeder
    driven as rank 0 of 2): init with device_id, broadcast on the side stream, event-ordered buffer reuse -- and the
    engine's outputs identical to the directly-fed run.  In a child process: the process group is global state."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "feed_nccl_selftest.py")], capture_output=True, text=True,
                       timeout=400, env=dict(os.environ, MASTER_PORT="29581"))
    assert r.returncode == 0 and "outputs identical to the direct feed" in r.stdout, (r.stdout[-400:], r.stderr[-800:])


# ---------------------------------------------------------------------------------------------------------------
# Calls of several blocks ("groups", xlating_batch_process_*_group): the results of G successive reference calls from
# one set of launches -- every client's phase is renormalised at each block end (xlating.c:73).
def _group_engine(fmt, max_input, gcap, clients, poly=None, m=None):
    eng = xl.BatchEngine(FS, fmt, max_input, group_blocks=gcap)
    if poly is not None:
        eng.set_option("polyphase", poly)
    if m is not None:
        eng.set_option("polyphase_m", m)
    oracles = {}
    for D, taps, fc in clients:
        oracles[eng.add_client(D, taps, fc)] = Oracle(D, taps, fc, FS, max_input)
    return eng, oracles


def _check_group(eng, oracles, fmt, x, G, variant, ids=None):
    eng.process_host_group(x, G, variant)
    eng.fetch()
    blocks = np.split(np.asarray(x), G)
    for cid, o in oracles.items():
        parts = [o.process(fmt, bl) for bl in blocks]
        if ids is not None and cid not in ids:
            continue
        want = np.concatenate(parts) if parts else np.zeros(0, np.complex64)
        got = eng.output(cid)
        assert eng.output_len(cid) == len(want), (cid, eng.output_len(cid), len(want))
        assert [eng.output_len_block(cid, g) for g in range(G)] == [len(p) for p in parts], cid
        if variant == "native":
            assert bits_equal(got, want), f"client {cid}"
        else:
            assert rel_err(got, want) <= REL_TOL, (cid, rel_err(got, want))


@pytest.mark.parametrize("variant", ["native", "optimized"])
def test_group_of_blocks_equals_successive_calls_direct(variant):
    """Direct kernels: mixed rates, G = 3 then 1 then 2 (the call shape changes: the look-ahead phase table is redone),
    block lengths that move the output grid, phases bit-exact at the end."""
    t48, t96, t101 = lpf(FS, 24000, 9600), lpf(FS, 48000, 19200), lpf(FS, 24000, 48000)
    clients = [(42, t48, -900000 + 91000 * c) for c in range(11)] + [(21, t96, 5000 * c) for c in range(5)] + \
              [(42, t101, -77777)]
    eng, oracles = _group_engine("cu8", 100002, 4, clients, poly=0)
    for k, (G, n) in enumerate(((3, 100002), (1, 100002), (2, 65536), (3, 100002), (3, 100002), (4, 2000))):
        x = siggen.xs_u8(4400 + k, G * n)
        _check_group(eng, oracles, "cu8", x, G, variant)
    if variant == "native":
        for cid, o in oracles.items():
            assert tuple(np.float32(v).tobytes() for v in eng.phase(cid)) == tuple(np.float32(v).tobytes() for v in o.phase)
    eng.close()


@pytest.mark.parametrize("m,inv,mix", [(128, 0, 1), (128, 3, 1), (256, 0, 1), (128, 5, 3), (256, 0, 3), (128, 6, 1)])
def test_group_of_blocks_polyphase(m, inv, mix, monkeypatch):
    """Forced polyphase path, G = 4 server-default blocks per call (108 segments at M = 128): every client vs the
    oracle's four successive calls; a native group in between (shared history and phases); ragged group.  mix = 1: the mix
    launch on the matrix cores with two-half operands (7 passes of 16 segments), 3: with float32 operands."""
    monkeypatch.setenv("XL_EXP_INV", str(inv))
    monkeypatch.setenv("XL_EXP_MIX", str(mix))
    t48 = lpf(FS, 24000, 9600)
    clients = [(42, t48, -900000 + 61000 * c) for c in range(30)]
    eng, oracles = _group_engine("cu8", 262144, 4, clients, poly=1, m=m)
    for k, (G, n, variant) in enumerate(((4, 262144, "optimized"), (4, 262144, "optimized"), (2, 262144, "native"),
                                        (3, 100002, "optimized"), (4, 262144, "optimized"), (1, 262144, "optimized"))):
        x = siggen.xs_u8(4500 + k, G * n)
        _check_group(eng, oracles, "cu8", x, G, variant)
    assert "polyphase: cls0 D42 T505 cols30" in eng.describe(), eng.describe()
    eng.close()


@pytest.mark.parametrize("m", [64, 128, 256])
@pytest.mark.parametrize("D", [50, 21, 37, 64])
def test_forward_groups_of_branches_cf32(D, m):
    """cf32 streams: the forward launch of a big call runs groups of adjacent branches per workgroup (four at M = 64 / 128, two at M = 256:
    one load per group and point, xl_polyphase.hip) -- D = 50 / 21 / 37 leave the last group partial (2 of 4 / 1 of 4, 1 of 2), 64 none; the first call
    reads below the clients' zero line and across history | block, a client joins between two calls (its own zero line inside the next
    call's window), the third call is ragged and short (the one-branch form again).  Calls of 8 blocks, every client per call against
    the oracle's eight successive calls (src/xlating.c:374-382: cf32 input needs no conversion)."""
    fs = 48000 * D
    taps = lpf(fs, 24000, 9600)
    nsamp = 98304
    gen = lambda seed, n: (siggen.xs_s16(seed, 2 * n).astype(np.float32) / np.float32(32768)).astype(np.float32)  # noqa: E731
    eng = xl.BatchEngine(fs, "cf32", 2 * nsamp, group_blocks=8)
    eng.set_option("polyphase", 1)
    eng.set_option("polyphase_m", m)
    oracles = {}
    for c in range(36):
        fc = int(-0.42 * fs + 0.024 * fs * c)
        oracles[eng.add_client(D, taps, fc)] = Oracle(D, taps, fc, fs, 2 * nsamp)
    for k, (G, n) in enumerate(((8, nsamp), (8, nsamp), (8, nsamp), (3, 50001), (8, nsamp))):
        if k == 2:
            oracles[eng.add_client(D, taps, 4321)] = Oracle(D, taps, 4321, fs, 2 * nsamp)
        x = gen(7100 + k, G * n)
        _check_group(eng, oracles, "cf32", x, G, "optimized")
        if k == 1:
            assert "polyphase: cls0 D%d " % D in eng.describe() and " M%d " % m in eng.describe(), eng.describe()
    eng.close()


def test_group_bench_shape_1024_clients_sampled():
    """The bench workload: 1024 x 48 kHz clients, 8 blocks per call, engine's own plan; 16 sampled clients."""
    t48 = lpf(FS, 24000, 9600)
    eng = xl.BatchEngine(FS, "cu8", 262144, group_blocks=8)
    fcs = [-984000 + 1920 * c for c in range(1024)]
    ids = [eng.add_client(42, t48, fc) for fc in fcs]
    sample = [0, 1, 63, 64, 127, 128, 255, 256, 500, 511, 512, 777, 1000, 1021, 1022, 1023]
    oracles = {ids[c]: Oracle(42, t48, fcs[c], FS, 262144) for c in sample}
    for k in range(3):
        x = siggen.xs_u8(4600 + k, 8 * 262144)
        _check_group(eng, oracles, "cu8", x, 8, "optimized")
    assert "polyphase: cls0 D42 T505 cols1024" in eng.describe(), eng.describe()
    _check_group(eng, oracles, "cu8", siggen.xs_u8(4610, 8 * 262144), 8, "native")
    eng.close()


def test_group_rejects_blocks_without_output():
    t48 = lpf(FS, 24000, 9600)
    eng = xl.BatchEngine(FS, "cu8", 262144, group_blocks=4)
    eng.add_client(42, t48, 1000)
    with pytest.raises(xl.XlatingError) as e:
        eng.process_host_group(np.zeros(4 * 40, np.uint8), 4, "native")  # 20 samples per block < D = 42
    assert e.value.code == -22
    with pytest.raises(xl.XlatingError):
        eng.process_host_group(np.zeros(5 * 1000, np.uint8), 5, "native")  # more blocks than the engine was built for
    eng.process_host_group(np.zeros(40, np.uint8), 1, "native")  # a single tiny block is fine
    eng.close()


# ---------------------------------------------------------------------------------------------------------------
# Clients that joined at different stream positions (dsp_worker.c:98-104: every client starts its own output grid).
def test_staggered_joins_merge_into_one_polyphase_class():
    """1024 x 48 kHz clients joining over 21 consecutive blocks (S mod D = 32: every join lands on another grid
    offset).  Once inside their own stream they all share ONE polyphase class (taps delayed by the grid offset, baked
    into the branch spectra); 16 sampled clients <= 1e-5 vs the oracle throughout, native bit-exact afterwards."""
    t48 = lpf(FS, 24000, 9600)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    fcs = [-984000 + 1920 * c for c in range(1024)]
    per = [49] * 20 + [44]
    sample_pos = {0, 48, 49, 100, 500, 979, 980, 1023}
    oracles = {}
    nxt = 0
    for k in range(24):
        if k < 21:
            for _ in range(per[k]):
                cid = eng.add_client(42, t48, fcs[nxt])
                if nxt in sample_pos or nxt % 128 == 5:
                    oracles[cid] = Oracle(42, t48, fcs[nxt], FS, 262144)
                nxt += 1
        x = siggen.xs_u8(4700 + k, 262144)
        check_clients(eng, oracles, "cu8", x, "optimized")
    assert nxt == 1024 and len(oracles) == 16
    d = eng.describe()
    assert "clients 1024 classes 21 " in d and "polyphase: cls0 D42 T505 cols1024 " in d and "cls1" not in d, d
    assert "optimized-mode direct: none" in d, d
    check_clients(eng, oracles, "cu8", siggen.xs_u8(4790, 262144), "native")
    check_clients(eng, oracles, "cu8", siggen.xs_u8(4791, 100002), "optimized")
    eng.close()


def test_many_classes_no_limit():
    """More than 48 distinct (D, T, grid offset) classes in one engine: blocks of 1009 samples (= 1 mod 42), a 48 kHz
    client joining before each of 42 blocks and a 96 kHz client before each of the first 21 -> 63 classes, every one
    with its own output grid.  (Round 1 failed every client with -E2BIG at the 49th class.)"""
    t48, t96 = lpf(FS, 24000, 9600), lpf(FS, 48000, 19200)
    eng = xl.BatchEngine(FS, "cu8", 20000)
    oracles = {}
    for k in range(44):
        if k < 42:
            fc = -500000 + 17000 * k
            oracles[eng.add_client(42, t48, fc)] = Oracle(42, t48, fc, FS, 20000)
        if k < 21:
            fc = 300000 + 9000 * k
            oracles[eng.add_client(21, t96, fc)] = Oracle(21, t96, fc, FS, 20000)
        check_clients(eng, oracles, "cu8", siggen.xs_u8(4800 + k, 2 * 1009), "native" if k % 2 else "optimized")
    d = eng.describe()
    assert "clients 63 classes 63 " in d, d
    eng.close()


def test_calls_on_different_streams_are_ordered():
    """Consecutive calls on different HIP streams (advice r1): the engine orders them itself."""
    import torch

    t48 = lpf(FS, 24000, 9600)
    eng = xl.BatchEngine(FS, "cu8", 262144)
    fcs = [-900000 + 28000 * c for c in range(40)]
    oracles = {eng.add_client(42, t48, fc): Oracle(42, t48, fc, FS, 262144) for fc in fcs}
    streams = [torch.cuda.Stream() for _ in range(3)]
    blocks = [siggen.xs_u8(4900 + k, 262144) for k in range(6)]
    dev = [torch.from_numpy(b).cuda() for b in blocks]
    torch.cuda.synchronize()
    wants = {cid: [o.process("cu8", b) for b in blocks] for cid, o in oracles.items()}
    for k in range(6):
        st = streams[k % 3]
        eng.process_device(dev[k].data_ptr(), 262144, "native", st.cuda_stream)
        if k in (2, 5):
            eng.fetch()
            for cid in oracles:
                assert bits_equal(eng.output(cid), wants[cid][k]), (k, cid)
    eng.close()


def test_set_option_and_unknown_option():
    eng = xl.BatchEngine(FS, "cu8", 262144)
    with pytest.raises(xl.XlatingError) as e:
        eng.set_option("no_such_option", 1)
    assert e.value.code == -2
    with pytest.raises(xl.XlatingError):
        eng.set_option("polyphase_m", 100)
    for name, bad in (("inverse_kernel", 1), ("inverse_kernel", 4), ("inverse_kernel", 7), ("mix_kernel", 0), ("mix_kernel", 2), ("nco_side_stream", 2)):
        with pytest.raises(xl.XlatingError):
            eng.set_option(name, bad)
    with pytest.raises(xl.XlatingError) as e:  # (round 1-4 tuning names are no options any more: XL_EXP_* at create)
        eng.set_option("riders", 0)
    assert e.value.code == -2
    eng.set_option("polyphase", 1)
    eng.set_option("polyphase_m", 256)
    t48 = lpf(FS, 24000, 9600)
    for c in range(5):
        eng.add_client(42, t48, 1000 * c)
    assert "polyphase: cls0 D42 T505 cols5 V244 M256" in eng.describe()
    eng.close()


@pytest.mark.parametrize("side,G", [(1, 1), (0, 1), (0, 4), (1, 3)])
@pytest.mark.parametrize("poly", [0, 1])
def test_nco_tabulation_side_stream_or_inside_the_launches(side, G, poly):
    """The next call's phase table comes either from the NCO role inside the launches or from xl_nco_chain_kernel on
    the side stream (default for calls of >= 2 blocks and for one-block polyphase calls): both forced here for both call shapes, native bit-exact incl.
    the committed phases, a shape change in between (the look-ahead table is dropped and redone)."""
    t48 = lpf(FS, 24000, 9600)
    clients = [(42, t48, -700000 + 47000 * c) for c in range(70)]
    eng, oracles = _group_engine("cu8", 100002, 4, clients, poly=poly)
    eng.set_option("nco_side_stream", side)
    for k, (g, n, variant) in enumerate(((G, 100002, "native"), (G, 100002, "optimized"), (G, 100002, "native"),
                                        (1 if G > 1 else 2, 65536, "native"), (G, 100002, "optimized"), (G, 100002, "native"))):
        x = siggen.xs_u8(5100 + k, g * n)
        _check_group(eng, oracles, "cu8", x, g, variant)
    for cid, o in oracles.items():
        assert tuple(np.float32(v).tobytes() for v in eng.phase(cid)) == tuple(np.float32(v).tobytes() for v in o.phase)
    eng.close()


@pytest.mark.parametrize("how", ["rank", "local", "loopback"])
def test_c_multi_host_single_gpu(how):
    """include/xlating_multi.h on the one GPU of this box (world 1 as a rank, and the one-process form with ngpus = 1):
    sharding call, feed of 4-block super-blocks from device memory, engine access, sync -- every client vs the oracle."""
    import torch

    t48 = lpf(FS, 24000, 9600)
    if how == "rank":
        m = xl.MultiHost(FS, "cu8", 100002, group_blocks=4, rank=0, world=1)
    elif how == "loopback":  # a one-rank RCCL communicator: the feed takes the broadcast path of world > 1
        m = xl.MultiHost(FS, "cu8", 100002, group_blocks=4, rank=0, world=1, uid=xl.MultiHost.unique_id())
    else:
        m = xl.MultiHost(FS, "cu8", 100002, group_blocks=4, ngpus=1)
    assert m.world == 1
    eng = m.engine(0)
    assert eng is not None and m.engine(1) is None
    oracles = {}
    for c in range(20):
        cid = m.add_client(c, 42, t48, -500000 + 50000 * c)
        oracles[cid] = Oracle(42, t48, -500000 + 50000 * c, FS, 100002)
    for k in range(3):
        x = siggen.xs_u8(5300 + k, 4 * 100002)
        d = torch.from_numpy(x).cuda()
        m.feed(d.data_ptr(), 100002, 4, "optimized" if k == 1 else "native")
        m.sync()
        eng.fetch()
        for cid, o in oracles.items():
            want = np.concatenate([o.process("cu8", bl) for bl in np.split(x, 4)])
            got = eng.output(cid)
            if k == 1:
                assert rel_err(got, want) <= REL_TOL
            else:
                assert bits_equal(got, want), (k, cid)
    m.close()


@pytest.mark.parametrize("loopback", [False, True], ids=["in-place", "loop-back communicator"])
def test_multi_feed_done_one_source_buffer_refilled_every_feed(loopback):
    """xlating_multi_feed_done / _feed_query / _feed_wait_on_stream (include/xlating_multi.h): a streaming host that owns
    ONE resident source buffer and refills it for every feed.  `loopback`: world 1 with an RCCL id -- a one-rank
    communicator, i.e. the broadcast path of world > 1 line for line (receive buffers, ready / free events, the source
    event behind the broadcast); otherwise the engine filters the source in place and the event sits behind its launches.
    Every client of every feed vs the oracle; a refill that came too early would corrupt a feed."""
    import torch

    t48 = lpf(FS, 24000, 9600)
    n, G = 100002, 4
    uid = xl.MultiHost.unique_id() if loopback else None
    m = xl.MultiHost(FS, "cu8", n, group_blocks=G, rank=0, world=1, uid=uid)
    eng = m.engine(0)
    oracles = {}
    for c in range(160):  # (>= 128 mature clients: optimized calls take the polyphase launches + side-stream chain)
        cid = m.add_client(c, 42, t48, -900000 + 11000 * c)
        if c % 9 == 0:
            oracles[cid] = Oracle(42, t48, -900000 + 11000 * c, FS, n)
    assert m.feed_query() is True  # nothing fed yet
    src = torch.empty(G * n, dtype=torch.uint8, device="cuda")
    pinned = torch.empty(G * n, dtype=torch.uint8).pin_memory()
    copy_stream = torch.cuda.Stream()
    feeds = []
    for k in range(6):
        x = siggen.xs_u8(5400 + k, G * n)
        feeds.append(x)
        if k % 2 == 0:   # host-side wait, then a blocking refill
            m.feed_done()
            assert m.feed_query() is True
            src.copy_(torch.from_numpy(x))
            torch.cuda.synchronize()
        else:            # stream-side wait: an asynchronous refill ordered behind the source event
            m.feed_wait_on_stream(copy_stream.cuda_stream)
            pinned.copy_(torch.from_numpy(x))
            with torch.cuda.stream(copy_stream):
                src.copy_(pinned, non_blocking=True)
            copy_stream.synchronize()
        m.feed(src.data_ptr(), n, G, "optimized" if k >= 2 else "native")
        if k in (2, 5):  # compare some feeds right away, let the others run back to back
            m.sync()
            eng.fetch()
            for cid, o in oracles.items():
                for xb in feeds:
                    want = np.concatenate([o.process("cu8", bl) for bl in np.split(xb, G)])
                got = eng.output(cid)
                assert rel_err(got, want) <= REL_TOL, (k, cid)
            feeds = []
    m.feed_done()
    m.close()


def test_multi_feed_timing_and_comm_count_loopback():
    """xlating_multi_feed_timing / _feed_timing_read / _comm_count (round 4: what bench.py prints for a multi-GPU run) on the
    one-rank loop-back communicator: the communicator counts 1 rank, every feed but the last two is harvested, the broadcasts
    take a plausible time and the hidden part never exceeds them; timing changes no result."""
    import torch

    t48 = lpf(FS, 24000, 9600)
    n, G = 262144, 4
    m = xl.MultiHost(FS, "cu8", n, group_blocks=G, rank=0, world=1, uid=xl.MultiHost.unique_id())
    assert m.comm_count() == 1
    eng = m.engine(0)
    oracles = {}
    for c in range(160):
        cid = m.add_client(c, 42, t48, -900000 + 11000 * c)
        if c % 40 == 0:
            oracles[cid] = Oracle(42, t48, -900000 + 11000 * c, FS, n)
    m.feed_timing(True)
    srcs = [torch.from_numpy(siggen.xs_u8(5500 + k, G * n)).cuda() for k in range(3)]
    feeds = 12
    for k in range(feeds):
        m.feed(srcs[k % 3].data_ptr(), n, G, "optimized")
    m.sync()
    cnt, bcast_ms, hidden_ms = m.feed_timing_read()
    assert cnt == feeds - 1, cnt  # (every feed that has a predecessor; the ring keeps the latest 14)
    assert 0.0 < bcast_ms / cnt < 5.0 and 0.0 <= hidden_ms <= bcast_ms * (1 + 1e-6), (cnt, bcast_ms, hidden_ms)
    eng.fetch()
    for cid, o in oracles.items():
        for k in range(feeds):
            want = np.concatenate([o.process("cu8", bl) for bl in np.split(srcs[k % 3].cpu().numpy(), G)])
        assert rel_err(eng.output(cid), want) <= REL_TOL, cid
    no_comm = xl.MultiHost(FS, "cu8", n, group_blocks=G, rank=0, world=1)
    assert no_comm.comm_count() == 0
    no_comm.close()
    m.close()


# ---------------------------------------------------------------------------------------------------------------
# The Q15 (cs16 output) family on the batched boundary (XL_MODE_Q15): exact integers.
@pytest.mark.parametrize("fmt", ["cu8", "cs8", "cs16"])
def test_q15_mode_bit_exact(fmt):
    """Every client's cs16 stream == the oracle's process_<fmt>_cs16 (xlating.c:92-140, 416-447) bit for bit: mixed
    rates, ragged blocks, a late joiner, a call of 3 blocks, many blocks so that the truncating phase recurrence drifts."""
    t48, t96, t101 = lpf(FS, 24000, 9600), lpf(FS, 48000, 19200), lpf(FS, 24000, 48000)
    maxin = 65536
    eng = xl.BatchEngine(FS, fmt, maxin, group_blocks=3)
    oracles = {}
    clients = [(42, t48, -700000 + 101000 * c) for c in range(13)] + [(21, t96, 9000 * c) for c in range(3)] + [(42, t101, 123456)]
    for D, taps, fc in clients:
        oracles[eng.add_client(D, taps, fc)] = Oracle(D, taps, fc, FS, maxin)
    gen = {"cu8": siggen.xs_u8, "cs8": siggen.xs_s8, "cs16": siggen.xs_s16}[fmt]
    for k, (G, n) in enumerate(((1, 65536), (1, 20002), (3, 30000), (1, 65536), (1, 2), (2, 65536), (1, 40000))):
        if k == 3:
            oracles[eng.add_client(42, t48, 31337)] = Oracle(42, t48, 31337, FS, maxin)
        x = gen(6000 + k, G * n)
        eng.process_host_group(x, G, "q15")
        eng.fetch()
        for cid, o in oracles.items():
            want = np.concatenate([o.process(fmt, bl, "cs16") for bl in np.split(x, G)])
            got = eng.output_cs16(cid)
            assert got.shape == want.shape and np.array_equal(got, want), (k, cid)
        with pytest.raises(xl.XlatingError):
            eng.output(next(iter(oracles)))  # float accessor after a Q15 call
    eng.close()


def test_q15_mode_rejected_for_cf32_engines():
    eng = xl.BatchEngine(FS, "cf32", 4096)
    eng.add_client(42, lpf(FS, 24000, 48000), 1000)
    with pytest.raises(xl.XlatingError) as e:
        eng.process_host(np.zeros(4096, np.float32), "q15")
    assert e.value.code == -22
    eng.close()


@pytest.mark.parametrize("ncalls", [1, 2, 3, 4])
@pytest.mark.parametrize("poly", [0, 1])
def test_chain_launch_covers_several_calls(ncalls, poly, monkeypatch):
    """One side-stream chain launch tabulates up to four calls ahead (rings of phase tables and phase buffers in the engine; the
    tuning knob XL_EXP_CHAIN_CALLS, read at create, makes it fewer).  Runs of equal calls long enough to consume whole launches, a shape change and a client joining while
    look-ahead tables are pending (both drop them), a native call in between (fused launches when the direct FIR is
    heavy): native outputs and the committed phases bit-exact, optimized within tolerance."""
    t48 = lpf(FS, 24000, 9600)
    clients = [(42, t48, -700000 + 47000 * c) for c in range(70)]
    monkeypatch.setenv("XL_EXP_CHAIN_CALLS", str(ncalls))
    eng, oracles = _group_engine("cu8", 100002, 4, clients, poly=poly)
    eng.set_option("nco_side_stream", 1)
    seq = [(4, 100002, "native")] * 5 + [(2, 65536, "native")] + [(4, 100002, "optimized")] * 3 + [(4, 100002, "native")] * 2
    for k, (g, n, variant) in enumerate(seq):
        if k == 8:  # a join while look-ahead tables are pending
            oracles[eng.add_client(42, t48, 333333)] = Oracle(42, t48, 333333, FS, 100002)
        x = siggen.xs_u8(5300 + k, g * n)
        _check_group(eng, oracles, "cu8", x, g, variant)
    for cid, o in oracles.items():
        assert tuple(np.float32(v).tobytes() for v in eng.phase(cid)) == tuple(np.float32(v).tobytes() for v in o.phase)
    eng.close()


def test_describe_after_a_change_keeps_the_latest_outputs():
    """Outputs stay valid until the next process call (xlating_batch.h) -- also across add_client / remove_client /
    set_option followed by describe(): with calls behind it a dirty engine describes the plan those calls ran with and
    says a re-plan is pending, instead of re-planning (which reassigns every client's output row)."""
    t48 = lpf(FS, 24000, 9600)
    clients = [(42, t48, -700000 + 47000 * c) for c in range(20)]
    eng, oracles = _group_engine("cu8", 100002, 1, clients, poly=0)
    first = eng.describe()  # no call yet: plans
    assert "clients 20" in first and "pending" not in first
    x = siggen.xs_u8(7100, 100002)
    eng.process_host(x, "native")
    want = {cid: o.process("cu8", x) for cid, o in oracles.items()}
    eng.remove_client(3)
    late = eng.add_client(42, t48, 31337)  # reuses row / id 3 in a naive re-plan
    d = eng.describe()
    assert "re-plan pending" in d, d
    eng.fetch()
    for cid in oracles:
        if cid != 3:
            assert bits_equal(eng.output(cid), want[cid]), cid
    ptr, n = eng.output_device(5)
    assert n == len(want[5]) and ptr
    oracles.pop(3)
    oracles[late] = Oracle(42, t48, 31337, FS, 100002)
    check_clients(eng, oracles, "cu8", siggen.xs_u8(7101, 100002), "native")
    assert "pending" not in eng.describe() and "clients 20" in eng.describe()
    eng.close()


@pytest.mark.parametrize("ncalls", [1, 2, 4])
def test_side_and_fused_calls_mixed_while_the_host_runs_ahead(ncalls, monkeypatch):
    """Calls whose NCO chain runs on the side stream (optimized, polyphase) alternate with calls that carry it inside their
    launches (native, heavy direct launches), enqueued back to back WITHOUT host synchronisation, with chain launches
    shorter than the table ring (XL_EXP_CHAIN_CALLS < 4): a chain launch must wait for the readers of every table slot
    it overwrites, whichever kind of call read it last.  Outputs of the final calls and the committed phases vs the oracle."""
    import torch

    t48 = lpf(FS, 24000, 9600)
    n, G = 100002, 4
    clients = [(42, t48, -900000 + 12000 * c) for c in range(150)]
    monkeypatch.setenv("XL_EXP_CHAIN_CALLS", str(ncalls))
    eng, oracles = _group_engine("cu8", n, G, clients)
    keep = sorted(oracles)[::15]
    seq = ["optimized", "optimized", "native", "optimized", "native", "native", "optimized", "optimized", "optimized", "native",
           "optimized", "optimized"]
    xs = [siggen.xs_u8(7200 + k, G * n) for k in range(len(seq))]
    dev = [torch.from_numpy(x).cuda() for x in xs]
    torch.cuda.synchronize()
    for k, variant in enumerate(seq):  # no sync in between: the host runs as far ahead as the queues allow
        eng.process_device_group(dev[k].data_ptr(), n, G, variant, "engine")
    eng.sync()
    eng.fetch()
    for cid in keep:
        o = oracles[cid]
        for x in xs:
            want = np.concatenate([o.process("cu8", bl) for bl in np.split(x, G)])
        assert rel_err(eng.output(cid), want) <= REL_TOL, cid
        assert tuple(np.float32(v).tobytes() for v in eng.phase(cid)) == tuple(np.float32(v).tobytes() for v in o.phase), cid
    eng.close()


# ---------------------------------------------------------------------------------------------------------------
# Full-population parity: EVERY client of a shape against a population of oracle filters run on the host cores
# (oracle/population.c: one reference-model filter per client, spread over pthreads).
def _engine_outputs(eng, ids):
    eng.fetch()
    return [eng.output(i) for i in ids]


@pytest.mark.parametrize("variant", ["native", "optimized", "optimized-lanes8-inverse", "optimized-cut32-inverse", "optimized-f32-mix"])
def test_group_bench_shape_1024_clients_all(variant, monkeypatch):
    """The headline shape (bench.py / BASELINE configs[3] on one GPU): 1024 x 48 kHz clients, 505 taps, calls of 8
    server-default blocks.  ALL 1024 clients x one whole 8-block call (1.07 G client-samples, 25.6 M outputs) against
    the oracle population, after a first call that loads every filter's history and phase: native bit for bit,
    optimized max|d| / max|y| <= 1e-5 per client (fixture semantics: test/test_xlating.c:24-61, test/utils.c:176-196) -- with the
    default launches (the inverse launch of this size: the LDS transform), with the 8-lane inverse kernel, and with float32 operands in
    the mix launch."""
    from pyoracle import population

    if variant.endswith("-lanes8-inverse"):
        monkeypatch.setenv("XL_EXP_INV", "5")
        variant = "optimized"
    if variant.endswith("-cut32-inverse"):
        monkeypatch.setenv("XL_EXP_INV", "6")
        variant = "optimized"
    if variant.endswith("-f32-mix"):  # float32 operands on the matrix cores: the all-float32 arithmetic of the path
        monkeypatch.setenv("XL_EXP_MIX", "3")
        variant = "optimized"
    t48 = lpf(FS, 24000, 9600)
    G, nb = 8, 262144
    fcs = [-984000 + 1920 * c for c in range(1024)]
    eng = xl.BatchEngine(FS, "cu8", nb, group_blocks=G)
    ids = [eng.add_client(42, t48, fc) for fc in fcs]
    x = siggen.xs_u8(8100, 2 * G * nb)
    for k in range(2):
        eng.process_host_group(x[k * G * nb:(k + 1) * G * nb], G, variant)
    got = _engine_outputs(eng, ids)
    if variant == "optimized":
        assert "polyphase: cls0 D42 T505 cols1024" in eng.describe(), eng.describe()
        assert ("mix=mf32" if os.environ.get("XL_EXP_MIX") == "3" else "mix=mfma") in eng.describe(), eng.describe()
        # (6912 tiles per launch: the size rule takes the LDS transform)
        assert {"5": "inv=lanes8", "6": "inv=cut32"}.get(os.environ.get("XL_EXP_INV"), "inv=lds") in eng.describe(), eng.describe()
    want = population(42, t48, fcs, FS, nb, "cu8", x, G, nwarm=G)
    worst = 0.0
    for c in range(1024):
        assert len(got[c]) == len(want[c]) == 24966, (c, len(got[c]), len(want[c]))
        if variant == "native":
            assert bits_equal(got[c], want[c]), c
        else:
            worst = max(worst, rel_err(got[c], want[c]))
    assert worst <= REL_TOL, worst
    eng.close()


def test_one_block_calls_on_the_engine_stream():
    """The reference's call granularity on the engine's own stream (XL_STREAM_ENGINE, what include/xlating_multi.h feeds):
    1024 clients, one 262144-byte block per call, 40 calls enqueued back to back without a host wait; the outputs of the calls that
    are looked at -- sampled clients after calls 9, 10 and 39, a native call and a host-path call in between -- match the oracle,
    and the committed phases are the oracle's bit for bit."""
    import torch

    t48 = lpf(FS, 24000, 9600)
    nb = 262144
    eng = xl.BatchEngine(FS, "cu8", nb)
    fcs = [-984000 + 1920 * c for c in range(1024)]
    ids = [eng.add_client(42, t48, fc) for fc in fcs]
    sample = [0, 1, 63, 64, 500, 777, 1022, 1023]
    oracles = {ids[c]: Oracle(42, t48, fcs[c], FS, nb) for c in sample}
    blocks = [siggen.xs_u8(8600 + k, nb) for k in range(6)]
    dev = [torch.from_numpy(x).cuda() for x in blocks]
    look = {9, 10, 20, 39}
    for k in range(40):
        x = blocks[k % 6]
        if k == 20:    # a native call in the middle of the run
            eng.process_device_group(dev[k % 6].data_ptr(), nb, 1, "native", "engine")
        elif k == 30:  # ... and a host-path call (the engine's plain stream)
            eng.process_host(x, "optimized")
        else:
            eng.process_device_group(dev[k % 6].data_ptr(), nb, 1, "optimized", "engine")
        want = {cid: o.process("cu8", x) for cid, o in oracles.items()}
        if k in look:
            eng.fetch()  # (xlating_batch_sync inside)
            for cid in oracles:
                got = eng.output(cid)
                if k == 20:
                    assert bits_equal(got, want[cid]), (k, cid)
                else:
                    assert rel_err(got, want[cid]) <= REL_TOL, (k, cid, rel_err(got, want[cid]))
    for cid, o in oracles.items():
        pr, pi = eng.phase(cid)
        assert (np.float32(pr), np.float32(pi)) == tuple(np.float32(v) for v in o.phase), cid
    assert "polyphase: cls0 D42 T505 cols1024" in eng.describe(), eng.describe()
    eng.close()


@pytest.mark.parametrize("variant,mix", [("native", 1), ("optimized", 1), ("optimized", 3)], ids=["native", "optimized", "optimized-f32-mix"])
def test_group_2048_clients_all(variant, mix, monkeypatch):
    """The shape the >= 50 % claim of DESIGN 6 rests on (the launches, not the recurrence, bound the call): 2048 x 48 kHz
    clients (16 column groups), one whole 8-block call after a warm-up call, ALL clients against the oracle population --
    native bit for bit, optimized <= 1e-5 per client (fixture semantics: test/test_xlating.c:24-61, test/utils.c:176-196)."""
    from pyoracle import population

    monkeypatch.setenv("XL_EXP_MIX", str(mix))
    t48 = lpf(FS, 24000, 9600)
    G, nb, n = 8, 262144, 2048
    fcs = [-984000 + 960 * c for c in range(n)]
    eng = xl.BatchEngine(FS, "cu8", nb, group_blocks=G)
    ids = [eng.add_client(42, t48, fc) for fc in fcs]
    x = siggen.xs_u8(8300, 2 * G * nb)
    for k in range(2):
        eng.process_host_group(x[k * G * nb:(k + 1) * G * nb], G, variant)
    got = _engine_outputs(eng, ids)
    if variant == "optimized":
        assert "polyphase: cls0 D42 T505 cols2048" in eng.describe() and ("mix=mf32" if mix == 3 else "mix=mfma") in eng.describe(), eng.describe()
    want = population(42, t48, fcs, FS, nb, "cu8", x, G, nwarm=G)
    worst = 0.0
    for c in range(n):
        assert len(got[c]) == len(want[c]) == 24966, (c, len(got[c]), len(want[c]))
        if variant == "native":
            assert bits_equal(got[c], want[c]), c
        else:
            worst = max(worst, rel_err(got[c], want[c]))
    assert worst <= REL_TOL, worst
    eng.close()


@pytest.mark.parametrize("mix", [1, 3], ids=["mfma-mix", "f32-mix"])
def test_group_4096_clients_sampled(mix, monkeypatch):
    """4096 x 48 kHz clients (32 column groups), 8 blocks per call, optimized: every 16th column and the first and last column
    of every group of 128 (and of 16) against oracle filters over two calls; a one-block call on the
    same engine afterwards (the reference's call granularity).  Also: the plan's choices at this size (inverse kernel by the size
    rule, the side kernel's CU reservation in rounds)."""
    monkeypatch.setenv("XL_EXP_MIX", str(mix))
    t48 = lpf(FS, 24000, 9600)
    G, nb, n = 8, 262144, 4096
    fcs = [-984000 + 480 * c for c in range(n)]
    eng = xl.BatchEngine(FS, "cu8", nb, group_blocks=G)
    ids = [eng.add_client(42, t48, fc) for fc in fcs]
    sample = sorted(set(range(0, n, 16)) | set(range(127, n, 128)) | set(range(15, n, 256)) | {n - 1})
    oracles = {ids[c]: Oracle(42, t48, fcs[c], FS, nb) for c in sample}
    for k in range(2):
        _check_group(eng, oracles, "cu8", siggen.xs_u8(8400 + k, G * nb), G, "optimized")
    assert "polyphase: cls0 D42 T505 cols4096" in eng.describe() and ("mix=mf32" if mix == 3 else "mix=mfma") in eng.describe(), eng.describe()
    # (27 648 tiles per launch: the 32 x 4 inverse kernel; the side-stream recurrence kernel's 64 workgroups run in two rounds on 32
    # reserved CUs -- one CU each would be a quarter of the chip; with the float32 mix a client weighs 1.5 x as much launch time and
    # the plan reserves none: xl_plan_rules.h)
    assert "inv=cut32" in eng.describe() and (mix == 3 or "side kernel: 32 CUs reserved" in eng.describe()), eng.describe()
    check_clients(eng, oracles, "cu8", siggen.xs_u8(8410, nb), "optimized")
    eng.close()


def test_chain_kernel_placement_does_not_change_the_phases(monkeypatch):
    """The side-stream recurrence kernel of a 4096-client plan under its three placements -- in two rounds on 32 reserved CUs (the rule),
    one CU per workgroup (XL_EXP_ROUNDS1, round 3's rule), no reservation (XL_EXP_NOMASK: the chain workgroups wait for whole free CUs)
    -- is the same arithmetic at different times: three engines fed the same 96 calls of 8 blocks must hold bit-identical phases for all
    4096 clients every 16 calls -- and bit-identical outputs --, whatever the look-ahead, the event order and the queueing did in between."""
    t48 = lpf(FS, 24000, 9600)
    G, nb, n = 8, 65536, 4096
    fcs = [-984000 + 480 * c for c in range(n)]
    engs = []
    x0 = siggen.xs_u8(8700, G * nb)
    for env in (None, "XL_EXP_ROUNDS1", "XL_EXP_NOMASK"):
        if env:
            monkeypatch.setenv(env, "1")
        e = xl.BatchEngine(FS, "cu8", nb, group_blocks=G)
        ids = [e.add_client(42, t48, fc) for fc in fcs]
        e.process_host_group(x0, G, "optimized")  # (the plan, and with it the reservation, is made by the first call)
        e.sync()
        if env:
            monkeypatch.delenv(env)
        engs.append((e, ids))
    d = [e.describe() for e, _ in engs]
    assert "side kernel: 32 CUs reserved" in d[0] and "side kernel: 64 CUs reserved" in d[1] and "side kernel:" not in d[2], d
    for k in range(1, 96):
        x = siggen.xs_u8(8700 + k, G * nb)
        for e, _ in engs:
            e.process_host_group(x, G, "optimized")
        if k % 16 == 15:
            ph = []
            for e, ids in engs:
                e.sync()
                ph.append(np.array([e.phase(i) for i in ids], dtype=np.float32))
            for other in (1, 2):
                bad = np.flatnonzero((ph[0].view(np.uint32) != ph[other].view(np.uint32)).any(axis=1))
                assert len(bad) == 0, (k, other, bad[:32])
    outs = []
    for e, ids in engs:  # ... and so are the outputs of the last call (sampled columns of the first, a middle and the last column group)
        e.fetch()
        outs.append([np.array(e.output(ids[c])) for c in (0, 77, 2047, 2048, 4000, 4095)])
    for other in (1, 2):
        for a, b_ in zip(outs[0], outs[other]):
            assert len(a) > 0 and bits_equal(a, b_), other
    for e, _ in engs:
        e.close()


def test_group_2304_clients_sampled_no_cu_reservation():
    """2304 x 48 kHz clients, 8 blocks per call, optimized: between 2049 and 3008 clients the side-stream recurrence kernel gets no CUs of
    its own (xl_plan_rules.h: its 36 workgroups take whole CUs as the launches' tails free them) -- every 16th column and the last one
    against oracle filters over three calls, the committed phases of the sampled clients bit for bit at the end."""
    t48 = lpf(FS, 24000, 9600)
    G, nb, n = 8, 262144, 2304
    fcs = [-984000 + 850 * c for c in range(n)]
    eng = xl.BatchEngine(FS, "cu8", nb, group_blocks=G)
    ids = [eng.add_client(42, t48, fc) for fc in fcs]
    sample = sorted(set(range(0, n, 16)) | {n - 1})
    oracles = {ids[c]: Oracle(42, t48, fcs[c], FS, nb) for c in sample}
    for k in range(3):
        _check_group(eng, oracles, "cu8", siggen.xs_u8(8600 + k, G * nb), G, "optimized")
    d = eng.describe()
    assert "polyphase: cls0 D42 T505 cols2304" in d and "inv=cut32" in d and "side kernel:" not in d, d
    for cid, o in oracles.items():
        pr, pi = eng.phase(cid)
        assert (np.float32(pr), np.float32(pi)) == tuple(np.float32(v) for v in o.phase), cid
    eng.close()


@pytest.mark.parametrize("nclients", [64, 256, 1024])
def test_config5_cf32_10msps_all_clients(nclients):
    """BASELINE config 5 at the client counts SURVEY 8(d) lists (N = 64, 256; N = 1 runs in
    test_config5_cf32_10msps_257_taps's family): cf32 input at 10 Msps, D = 100, 257 explicit taps, S = 131072; every
    client of two consecutive blocks vs the oracle population, PER BLOCK: native bit-exact and optimized <= 1e-5 -- the optimized
    calls take the polyphase path with the mix on the matrix cores, by default on two-half operands (13 k-blocks of 8 branches, the
    spectra scaled per segment: round 6), on request (option mix_kernel = 3) on float32 operands; 1024 clients = the count
    bench.py's config-5 entry is quoted on."""
    from pyoracle import population

    taps = siggen.hamming_sinc(257, 0.004)
    nsamp = 131072
    fcs = [-4900000 + (9800000 // nclients) * c for c in range(nclients)]
    x = np.concatenate([siggen.sin_f32(0, 2 * nsamp), (siggen.xs_s16(91, 2 * nsamp).astype(np.float32) / np.float32(32768))]).astype(np.float32)
    want0 = population(100, taps, fcs, 10000000, 2 * nsamp, "cf32", x[:2 * nsamp], 1)
    want1 = population(100, taps, fcs, 10000000, 2 * nsamp, "cf32", x, 1, nwarm=1)
    for variant, mix in (("native", 0), ("optimized", 0), ("optimized", 3)):
        eng = xl.BatchEngine(10000000, "cf32", 2 * nsamp)
        if mix:
            eng.set_option("mix_kernel", mix)
        ids = [eng.add_client(100, taps, fc) for fc in fcs]
        for k, want in enumerate((want0, want1)):
            eng.process_host(x[k * 2 * nsamp:(k + 1) * 2 * nsamp], variant)
            if variant == "optimized" and k == 1:  # (block 0: the clients are inside their zero history -- a class of its own)
                d = eng.describe()
                assert "polyphase: cls0 D100 T257 cols%d " % nclients in d and ("mix=mf32" if mix == 3 else "mix=mfma") in d, d
                assert " V62 M64 " in d, d  # (more than 64 branches, 3 taps per branch: 64-point transforms by the size rule)
                if mix != 3:  # (a plan with a wide two-half class reserves no CUs for the side-stream kernel: xl_batch.cpp)
                    assert "CUs reserved" not in d, d
            got = _engine_outputs(eng, ids)
            for c in range(nclients):
                assert len(got[c]) == len(want[c]), c
                if variant == "native":
                    assert bits_equal(got[c], want[c]), (k, c)
                else:
                    assert rel_err(got[c], want[c]) <= REL_TOL, (variant, mix, k, c, rel_err(got[c], want[c]))
        eng.close()


def _cf32_adversarial_blocks(nsamp):
    """8 consecutive cf32 blocks (interleaved I,Q float32) whose levels and shapes differ wildly INSIDE one call: what the per-segment
    scales of the two-half mix are for."""
    rng = np.random.default_rng(606)
    n = 2 * nsamp
    noise = lambda a: (rng.standard_normal(n) * a).astype(np.float32)  # noqa: E731
    loud = noise(0.3)
    quiet = noise(0.3e-4)                                   # 10^4 x quieter than the block before it
    square = np.where((np.arange(n) // 2) % 200 < 100, 1.0, -1.0).astype(np.float32)   # full-scale square wave on I and Q
    tiny = noise(1e-6)
    impulse = np.zeros(n, np.float32)
    impulse[2 * 4321] = 1.0                                  # a lone impulse in an all-zero block
    zeros = np.zeros(n, np.float32)
    huge = noise(3e4)                                        # far above any integer format's range
    ramp = (np.linspace(-1.0, 1.0, n) ** 3).astype(np.float32) * np.float32(1e-3)
    return [loud, quiet, square, tiny, impulse, zeros, huge, ramp]


@pytest.mark.parametrize("shape", ["config5_d100", "d42_505_taps"])
def test_cf32_two_half_mix_adversarial_levels_in_one_call(shape):
    """cf32 input has no a-priori bound, so the two-half matrix-core mix scales the shared spectra per SEGMENT (the forward launch finds
    each segment's largest component; a row scale factors out of the per-bin matrix product exactly).  ONE 8-block call whose blocks are:
    loud noise, noise 10^4 x quieter, a full-scale square wave, 1e-6-amplitude noise, a lone impulse, zeros, 3e4-amplitude noise, a
    small cubic ramp -- every client, PER BLOCK, against the oracle (the reference's float32 direct form, src/xlating.c:52-83), on both
    the default plan (two-half operands) and the float32-operand option; then the same eight blocks again as eight one-block calls
    (another segmentation of the same stream).
    The bar: max|d| <= 1e-5 x the block's own max|y| -- for every block whose neighbours are not more than 100 x louder (the loud block,
    the quiet block BEHIND it as far as its own scale reaches, the square wave, the impulse, the huge noise, the ramp).  An overlap-save
    evaluation leaves rounding of the SEGMENT's scale -- about 12 600 (D = 100) / 10 200 (D = 42) input samples -- in every output of a
    segment, with two-half AND with float32 operands alike (float32 epsilon x the segment's largest spectrum value); the reference's
    direct form rounds relative to a tap window.  So the last outputs of a block that sit in one segment with the first samples of a
    >= 100 x louder NEXT block (here: the quiet noise before the full-scale square wave, 2-4e-5 of the quiet block's scale measured with
    either mix; the 1e-6 noise before the impulse; the all-zero block, whose reference outputs are exact zeros, before the 3e4 noise) are
    held to 1e-5 of the LOUDER neighbour's output scale instead.  DESIGN.md (numerics) states this as the path's one deviation in kind
    from the reference's error behaviour."""
    nsamp = 131072
    if shape == "config5_d100":
        fs, D, taps = 10000000, 100, siggen.hamming_sinc(257, 0.004)
    else:
        fs, D, taps = FS, 42, lpf(FS, 24000, 9600)
    blocks = _cf32_adversarial_blocks(nsamp)
    nclients = 40
    fcs = [int(-0.45 * fs + (0.9 * fs / nclients) * c) for c in range(nclients)]
    oracles = [Oracle(D, taps, fc, fs, 8 * nsamp) for fc in fcs]
    warm = (siggen.xs_s16(17, 2 * nsamp).astype(np.float32) / np.float32(32768)).astype(np.float32)
    want_warm = [o.process("cf32", warm) for o in oracles]
    want = [[o.process("cf32", xb) for xb in blocks] for o in oracles]          # first pass over the eight blocks
    want2 = [[o.process("cf32", xb) for xb in blocks] for o in oracles]         # second pass
    for mix in (0, 3):
        eng = xl.BatchEngine(fs, "cf32", 2 * nsamp, group_blocks=8)
        if mix:
            eng.set_option("mix_kernel", mix)
        ids = [eng.add_client(D, taps, fc) for fc in fcs]
        eng.process_host(warm, "optimized")  # (the clients leave their zero history)
        eng.fetch()
        for c in range(nclients):
            assert rel_err(eng.output(ids[c]), want_warm[c]) <= REL_TOL
        eng.process_host_group(np.concatenate(blocks), 8, "optimized")
        d = eng.describe()
        assert "polyphase: cls0 D%d " % D in d and ("mix=mf32" if mix == 3 else "mix=mfma") in d, d
        eng.fetch()
        worst = 0.0
        for c in range(nclients):
            got = eng.output(ids[c])
            lens = [eng.output_len_block(ids[c], g) for g in range(8)]
            assert lens == [len(w) for w in want[c]], (c, lens)
            off = 0
            scale = [float(np.abs(wb).max()) for wb in want[c]]
            for g in range(8):
                gb, wb = got[off:off + lens[g]], want[c][g]
                off += lens[g]
                near = max(scale[max(g - 1, 0)], scale[min(g + 1, 7)])
                ref = scale[g] if near <= 100.0 * scale[g] else near  # (see the docstring: a >= 100 x louder neighbour sets the scale)
                e = float(np.abs(gb.astype(np.complex128) - wb).max()) / ref
                worst = max(worst, e)
                assert e <= REL_TOL, (shape, mix, c, g, e, "own scale" if ref == scale[g] else "neighbour's scale")
        for g in range(8):  # the same blocks, one per call
            eng.process_host(blocks[g], "optimized")
            eng.fetch()
            for c in range(nclients):
                wb = want2[c][g]
                gb = eng.output(ids[c])
                assert len(gb) == len(wb)
                # (one block per call: a segment never reaches into the NEXT block -- the call ends with the block -- but its first one
                # starts in the previous block's tail, whose samples the history holds)
                sc = [float(np.abs(w).max()) for w in want2[c]]
                prev = sc[g - 1] if g > 0 else float(np.abs(want[c][7]).max())
                ref = sc[g] if prev <= 100.0 * sc[g] else prev
                e = float(np.abs(gb.astype(np.complex128) - wb).max()) / ref
                assert e <= REL_TOL, (shape, mix, c, g, "one block per call", e)
        eng.close()


def test_plain_process_ignores_tuning_variables():
    """With XL_EXP_* set but XL_TESTING unset (a production process), describe() equals the default plan; with XL_TESTING=1 the same
    variables force the plan (what the tests above rely on)."""
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import siggen, sdr_server_amd as xl\n"
            "t = xl.create_low_pass_filter(1.0, 2016000, 24000, 9600)[1]\n"
            "e = xl.BatchEngine(2016000, 'cu8', 262144, group_blocks=2)\n"
            "[e.add_client(42, t, -800000 + 41000 * c) for c in range(40)]\n"
            "[e.process_host_group(siggen.xs_u8(9 + k, 2 * 262144), 2, 'optimized') for k in range(2)]\n"
            "print('PLAN', e.describe())\n" % (ROOT, os.path.join(ROOT, "tests")))
    base = {k: v for k, v in os.environ.items() if not k.startswith("XL_")}
    knobs = {"XL_EXP_POLY": "0", "XL_EXP_MIX": "3", "XL_EXP_POLY_M": "256", "XL_EXP_NOMASK": "1", "XL_EXP_RESERVE": "1"}

    def plan(env):
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1000:]
        return next(l for l in r.stdout.splitlines() if l.startswith("PLAN")), r.stderr

    default, _ = plan(base)
    stray, err = plan(dict(base, **knobs))
    forced, _ = plan(dict(base, XL_TESTING="1", **knobs))
    assert "polyphase: cls0 D42 T505 cols40 " in default and "mix=mfma" in default, default
    assert stray == default and "tuning variables are set but ignored" in err, (stray, err[-300:])
    assert "polyphase: none" in forced, forced


def test_config5_fresh_engines_tone_block_every_client_repeated():
    """The stress that exposed the unshipped k-block-major wide mix (tools/experiments/mix_wide_kmajor/: wrong sums in a few workgroups
    per launch, timing dependent, 50-100 % of fresh engines): BASELINE config 5 at 256 clients, the first block of a pure-tone stream --
    most clients sit in the tone's stopband, so a single tile of a single bin off by 2^-11 of its value shows as 1e-4 of max|y| --, a
    NEW engine per round right behind another engine's life in the same process, every client of every round within 1e-5.  The shipped
    kernels pass it 12 of 12 (profiles/r06_mix_wide_kmajor_wrong_sums.txt (10)); six rounds here."""
    from pyoracle import population

    taps = siggen.hamming_sinc(257, 0.004)
    nsamp, n = 131072, 256
    fcs = [-4900000 + (9800000 // n) * c for c in range(n)]
    x = siggen.sin_f32(0, 2 * nsamp).astype(np.float32)
    want = population(100, taps, fcs, 10000000, 2 * nsamp, "cf32", x, 1)
    for rnd in range(6):
        other = xl.BatchEngine(10000000, "cf32", 2 * nsamp)  # (another engine's allocations and launches first)
        for fc in fcs[:64]:
            other.add_client(100, taps, fc)
        other.process_host(x, "native" if rnd % 2 == 0 else "optimized")
        other.close()
        eng = xl.BatchEngine(10000000, "cf32", 2 * nsamp)
        ids = [eng.add_client(100, taps, fc) for fc in fcs]
        eng.process_host(x, "optimized")
        assert "mix=mfma" in eng.describe(), eng.describe()
        got = _engine_outputs(eng, ids)
        bad = [(c, rel_err(got[c], want[c])) for c in range(n) if rel_err(got[c], want[c]) > REL_TOL]
        assert not bad, (rnd, len(bad), bad[:8])
        eng.close()


def test_expected_clients_reserves_the_side_kernels_cus_once():
    """Option "expected_clients": the CUs of the side-stream recurrence kernel are reserved for the announced population at the
    first plan, so joins up to it never re-create the CU-masked streams (25 ms each time the count crosses a multiple of 512
    otherwise).  Same results either way (oracle, clients on both sides of the crossing)."""
    t48 = lpf(FS, 24000, 9600)
    nb = 131072
    reserved = {}
    for expect in (0, 1024):
        eng = xl.BatchEngine(FS, "cu8", nb, group_blocks=2)
        if expect:
            eng.set_option("expected_clients", expect)
        fcs = [-984000 + 1920 * c for c in range(700)]
        cids = [eng.add_client(42, t48, fc) for fc in fcs[:500]]
        watch = {cids[i]: Oracle(42, t48, fcs[i], FS, nb) for i in (0, 257, 499)}
        x0, x1 = siggen.xs_u8(4100, 2 * nb), siggen.xs_u8(4101, 2 * nb)
        _check_group(eng, watch, "cu8", x0, 2, "optimized")
        d0 = eng.describe()
        cids += [eng.add_client(42, t48, fc) for fc in fcs[500:]]  # 500 -> 700 clients: past 512
        watch[cids[650]] = Oracle(42, t48, fcs[650], FS, nb)
        _check_group(eng, watch, "cu8", x1, 2, "optimized")
        d1 = eng.describe()
        reserved[expect] = (d0.split("side kernel: ")[1].split()[0], d1.split("side kernel: ")[1].split()[0])
        for o in watch.values():
            o.close()
        eng.close()
    assert reserved[0] == ("8", "16"), reserved     # by the clients joined so far: one CU per XCD per 512 clients
    assert reserved[1024] == ("16", "16"), reserved  # announced: reserved once
    eng = xl.BatchEngine(FS, "cu8", nb)
    with pytest.raises(Exception):
        eng.set_option("expected_clients", -1)
    eng.close()


def test_churn_one_join_and_one_leave_per_block():
    """Incremental re-planning under churn: 1024 x 48 kHz clients, and for 200 blocks one client joins AND one leaves before
    every block (dsp_worker_start / dsp_worker_destroy while the stream runs, src/dsp_worker.c:90-108, 172-197).  A joiner
    spends its first block in a direct class of its own, then takes a column of the polyphase class -- the column a leaver
    freed when there is one -- and only that column's branch spectra are computed.  Checked against the oracle: six clients
    that stay for the whole run, every joiner over its first six blocks (alone, merged, settled), recycled client ids, and a
    native block every 16th (bit-exact; it rebuilds the all-clients launch set after the churn)."""
    t48 = lpf(FS, 24000, 9600)
    nb = 262144
    eng = xl.BatchEngine(FS, "cu8", nb)
    fcs = {eng.add_client(42, t48, -984000 + 1920 * c): -984000 + 1920 * c for c in range(1024)}
    stay = [0, 1, 511, 640, 1000, 1023]
    oracles = {cid: Oracle(42, t48, fcs[cid], FS, nb) for cid in stay}
    watch = {}  # cid -> blocks left to watch
    rng = np.random.default_rng(11)
    leavable = [cid for cid in fcs if cid not in stay]
    for k in range(200):
        gone = leavable.pop(int(rng.integers(len(leavable))))
        eng.remove_client(gone)
        if gone in oracles:
            oracles.pop(gone).close()
            watch.pop(gone, None)
        fc = int(rng.integers(-980000, 980000))
        cid = eng.add_client(42, t48, fc)  # (reuses the id that just left, half of the time)
        oracles[cid] = Oracle(42, t48, fc, FS, nb)
        watch[cid] = 6
        leavable.append(cid)
        x = siggen.xs_u8(9300 + k, nb if k % 5 else 200004)
        variant = "native" if k % 16 == 15 else "optimized"
        check_clients(eng, oracles, "cu8", x, variant)
        for c in list(watch):
            watch[c] -= 1
            if watch[c] == 0:
                del watch[c]
                oracles.pop(c).close()
    check_clients(eng, oracles, "cu8", siggen.xs_u8(9999, nb), "optimized")  # (the last joiner merges)
    d = eng.describe()
    assert "clients 1024 " in d and "polyphase: cls0 D42 T505 cols1024" in d and "cls1" not in d, d
    eng.close()
