"""CPU tests (-m "not gpu") of the drop-in boundary: the C-ABI library loads, exports every symbol the headers
declare, refuses to run without a GPU (no CPU fallback), and its host-side tap designer matches the reference."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import sdr_server_amd as xl
from conftest import ROOT, bits_equal, load_live, trunc1e4
import scenarios


def _declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([a-z_0-9]+)\s*\(", src)
    return sorted({n for n in names if n.startswith(("process_", "create_", "destroy_", "xlating_"))})


def test_library_exists_and_loads():
    assert os.path.exists(xl.library_path()), "run __graft_entry__.build() first"
    assert xl.simd_status() == "HIP gfx950"


@pytest.mark.parametrize("header", ["xlating.h", "lpf.h", "xlating_batch.h"])
def test_every_declared_symbol_is_exported(header):
    L = xl.lib()
    decl = _declared_functions(header)
    assert decl, header
    for name in decl:
        assert hasattr(L, name), f"{name} declared in include/{header} but not exported"
    if header == "xlating.h":
        # 1 create + 12 process + destroy (reference src/xlating.h:10-38) + 2 cf32-input extensions
        assert len([n for n in decl if n.startswith("process_")]) == 14
        C.c_char_p.in_dll(L, "SIMD_STATUS")


@pytest.mark.parametrize("header,libname", [("xlating_sinks.h", "hip"), ("xlating_wire.h", "hip"), ("xlating_multi.h", "multi")])
def test_every_declared_symbol_is_defined_nm(header, libname):
    """The remaining headers, by symbol table (libxlating_multi.so links librccl: not dlopen'ed on a GPU-less box)."""
    path = xl.library_path() if libname == "hip" else xl.multi_library_path()
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    defined = {l.split()[-1] for l in out.splitlines() if l.strip()}
    decl = _declared_functions(header)
    assert decl, header
    for name in decl:
        assert name in defined, f"{name} declared in include/{header} but not defined in {os.path.basename(path)}"
    if libname == "multi":
        assert set(xl.MULTI_SYMBOLS) <= defined


def test_exported_list_matches_nm():
    out = subprocess.run(["nm", "-D", "--defined-only", xl.library_path()], capture_output=True, text=True).stdout
    defined = {l.split()[-1] for l in out.splitlines() if l.strip()}
    for s in xl.EXPORTED_SYMBOLS:
        assert s in defined, s


def test_no_oracle_linked_into_product():
    out = subprocess.run(["nm", "-D", xl.library_path()], capture_output=True, text=True).stdout
    assert "orc_" not in out
    ldd = subprocess.run(["ldd", xl.library_path()], capture_output=True, text=True).stdout
    assert "liboracle" not in ldd and "libref" not in ldd


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="GPU present: covered by the gpu tests")
def test_fails_loudly_without_gpu(capfd):
    """No CPU arithmetic path: create returns -ENODEV and logs a <3> line (reference logging convention)."""
    with pytest.raises(xl.XlatingError) as e:
        xl.XlatingFilter(5, np.ones(57, np.float32), -12000, 48000, 2000)
    assert e.value.code == -19
    assert "<3>" in capfd.readouterr().err
    with pytest.raises(xl.XlatingError) as e:
        xl.BatchEngine(2016000, "cu8", 262144)
    assert e.value.code == -19


def test_create_rejects_empty_taps_without_consuming():
    """xlating.c:496-498: taps_len == 0 -> -1 (checked before any device work)"""
    h = C.c_void_p()
    buf = (C.c_float * 4)()
    assert xl.lib().create_frequency_xlating_filter(5, buf, 0, 0, 48000, 2000, C.byref(h)) == -1
    xl.lib().destroy_xlating(None)  # NULL-safe (xlating.c:585-587)


# ---- host-side tap designer (include/lpf.h) -- runs on the CPU by design (one-time, O(T)) ---------------------


def test_lpf_golden(ref_vectors):
    """test/test_lpf.c:25-39"""
    exp = ref_vectors["test_lpf.c"]["test_lowpassTaps"]["expected_taps"]
    code, taps = xl.create_low_pass_filter(1.0, 8000, 1750, 500)
    assert code == 0 and taps.size == 39
    assert np.array_equal(trunc1e4(exp), trunc1e4(taps))


@pytest.mark.parametrize("args", [(0, 1750, 500), (8000, 5000, 500), (8000, 1750, 0)])
def test_lpf_bounds(args):
    """test/test_lpf.c:7-23"""
    code, taps = xl.create_low_pass_filter(1.0, *args)
    assert code == -1 and taps is None


@pytest.mark.parametrize("sc", [s for s in scenarios.SCENARIOS if s["taps"][0] == "lpf"], ids=lambda s: s["name"])
def test_lpf_bit_exact_vs_committed_reference_taps(sc):
    code, taps = xl.create_low_pass_filter(1.0, sc["fs"], sc["taps"][1], sc["taps"][2])
    assert code == 0
    assert bits_equal(taps, load_live(sc["name"])["taps"])


# ---- host-side tap preparation (csrc/xl_taps.c; xlating.c:524-549) vs the oracle restatement ------------------


@pytest.mark.parametrize("sc", scenarios.SCENARIOS, ids=lambda s: s["name"])
def test_prepare_taps_bit_exact_vs_oracle(sc):
    from pyoracle import Oracle

    taps = scenarios.make_taps(sc, lpf=lambda *a: xl.create_low_pass_filter(*a)[1])
    T = taps.size
    rt = np.zeros(2 * T, np.float32)
    rq = np.zeros(2 * T, np.int16)
    inc = np.zeros(2, np.float32)
    qinc = np.zeros(2, np.int16)
    fn = xl.lib().xl_prepare_taps
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = None
    fn(taps.ctypes.data, T, sc["fc"], sc["fs"], sc["D"], rt.ctypes.data, rq.ctypes.data, inc.ctypes.data, qinc.ctypes.data)
    o = Oracle(sc["D"], taps, sc["fc"], sc["fs"], sc["max_input"])
    assert bits_equal(rt.view(np.complex64), o.rtaps)
    assert np.array_equal(rq.reshape(-1, 2), o.rtaps_q15)
    assert bits_equal(inc, np.array(o.phase_incr, np.float32))
    o.close()
