"""oracle/pyoracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes bindings for
  * oracle/liboracle.so          the repo's own CPU restatement (xlating_oracle.c)
  * oracle/_ref/libref_*.so      the UNMODIFIED reference compiled by oracle/Makefile
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

_c_float_p = C.POINTER(C.c_float)
_c_i16_p = C.POINTER(C.c_int16)


def build(ref=True):
    """(Re)build liboracle.so and, when /root/reference is present, oracle/_ref/."""
    subprocess.run(["make", "-C", HERE, "liboracle.so"], check=True, capture_output=True)
    if ref:
        subprocess.run(["make", "-C", HERE, "ref"], check=True, capture_output=True)


def _load(path):
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    return C.CDLL(path)


# --------------------------------------------------------------------------- restatement


class Oracle:
    """The repo's CPU restatement.  One instance == one filter (like `xlating *`)."""

    _lib = None
    IN_FMTS = {"cu8": (np.uint8, C.c_uint8), "cs8": (np.int8, C.c_int8), "cs16": (np.int16, C.c_int16),
               "cf32": (np.float32, C.c_float)}

    @classmethod
    def lib(cls):
        if cls._lib is None:
            p = os.path.join(HERE, "liboracle.so")
            if not os.path.exists(p):
                build(ref=False)
            L = _load(p)
            L.orc_lpf_design.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_c_float_p), C.POINTER(C.c_size_t)]
            L.orc_lpf_design.restype = C.c_int
            L.orc_lpf_ntaps.argtypes = [C.c_uint32, C.c_uint32]
            L.orc_lpf_ntaps.restype = C.c_int
            L.orc_xlating_create.argtypes = [C.c_uint32, _c_float_p, C.c_size_t, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
            L.orc_xlating_create.restype = C.c_int
            L.orc_xlating_destroy.argtypes = [C.c_void_p]
            L.orc_xlating_set_sum_mode.argtypes = [C.c_void_p, C.c_int]
            L.orc_xlating_set_renorm.argtypes = [C.c_void_p, C.c_int]
            L.orc_xlating_set_fma_step.argtypes = [C.c_void_p, C.c_int]
            L.orc_xlating_history.argtypes = [C.c_void_p]
            L.orc_xlating_history.restype = C.c_size_t
            L.orc_xlating_taps_len.argtypes = [C.c_void_p]
            L.orc_xlating_taps_len.restype = C.c_size_t
            L.orc_xlating_rtaps.argtypes = [C.c_void_p]
            L.orc_xlating_rtaps.restype = _c_float_p
            L.orc_xlating_rtaps_q15.argtypes = [C.c_void_p]
            L.orc_xlating_rtaps_q15.restype = _c_i16_p
            for n in ("orc_xlating_phase", "orc_xlating_phase_incr"):
                getattr(L, n).argtypes = [C.c_void_p, _c_float_p, _c_float_p]
            L.orc_xlating_phase_q15.argtypes = [C.c_void_p, _c_i16_p, _c_i16_p]
            L.orc_xlating_skip_calls_cf32.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
            L.orc_xlating_skip_calls_cf32.restype = None
            L.orc_hypotf_via_double.argtypes = [C.c_float, C.c_float]
            L.orc_hypotf_via_double.restype = C.c_float
            for fmt, (_, ct) in cls.IN_FMTS.items():
                fn = getattr(L, f"orc_process_{fmt}_cf32")
                fn.argtypes = [C.POINTER(ct), C.c_size_t, C.POINTER(_c_float_p), C.POINTER(C.c_size_t), C.c_void_p]
                if fmt != "cf32":
                    fn = getattr(L, f"orc_process_{fmt}_cs16")
                    fn.argtypes = [C.POINTER(ct), C.c_size_t, C.POINTER(_c_i16_p), C.POINTER(C.c_size_t), C.c_void_p]
            cls._lib = L
        return cls._lib

    @classmethod
    def lpf(cls, gain, fs, cutoff, tw):
        """-> (code, float32 taps or None)   [lpf.c:53]"""
        L = cls.lib()
        p = _c_float_p()
        n = C.c_size_t(0)
        code = L.orc_lpf_design(gain, fs, cutoff, tw, C.byref(p), C.byref(n))
        if code != 0:
            return code, None
        taps = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
        C.CDLL(None).free(p)
        return 0, taps

    def __init__(self, decimation, taps, center_freq, sampling_freq, max_input, sum_mode=0, renorm=True, fma_step=False):
        L = self.lib()
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        h = C.c_void_p()
        code = L.orc_xlating_create(decimation, taps.ctypes.data_as(_c_float_p), taps.size, center_freq, sampling_freq, max_input, C.byref(h))
        if code != 0:
            raise RuntimeError(f"orc_xlating_create -> {code}")
        self.h = h
        self.D = decimation
        if sum_mode:
            L.orc_xlating_set_sum_mode(h, sum_mode)
        if not renorm:  # process_optimized_* of the reference's x86 AVX build (xlating.c:338-339)
            L.orc_xlating_set_renorm(h, 0)
        if fma_step:    # ... of a build with FMA enabled
            L.orc_xlating_set_fma_step(h, 1)

    def close(self):
        if self.h:
            self.lib().orc_xlating_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process(self, fmt, x, out="cf32"):
        """x: 1-D array of scalar elements (I,Q interleaved).  Returns complex64[K] or int16[K,2]."""
        L = self.lib()
        npdt, ct = self.IN_FMTS[fmt]
        x = np.ascontiguousarray(x, dtype=npdt)
        n = C.c_size_t(0)
        if out == "cf32":
            p = _c_float_p()
            getattr(L, f"orc_process_{fmt}_cf32")(x.ctypes.data_as(C.POINTER(ct)), x.size, C.byref(p), C.byref(n), self.h)
            if n.value == 0:
                return np.zeros(0, np.complex64)
            return np.ctypeslib.as_array(p, shape=(2 * n.value,)).copy().view(np.complex64)
        p = _c_i16_p()
        getattr(L, f"orc_process_{fmt}_cs16")(x.ctypes.data_as(C.POINTER(ct)), x.size, C.byref(p), C.byref(n), self.h)
        if n.value == 0:
            return np.zeros((0, 2), np.int16)
        return np.ctypeslib.as_array(p, shape=(n.value, 2)).copy()

    def skip_calls(self, nsamples, ncalls):
        """Advance the stream state (phase, history counter) over ncalls cf32-family calls of nsamples complex samples
        without filtering; feed one real block afterwards before comparing outputs."""
        self.lib().orc_xlating_skip_calls_cf32(self.h, nsamples, ncalls)

    @property
    def history(self):
        return self.lib().orc_xlating_history(self.h)

    @property
    def phase(self):
        a, b = C.c_float(), C.c_float()
        self.lib().orc_xlating_phase(self.h, C.byref(a), C.byref(b))
        return np.float32(a.value), np.float32(b.value)

    @property
    def phase_incr(self):
        a, b = C.c_float(), C.c_float()
        self.lib().orc_xlating_phase_incr(self.h, C.byref(a), C.byref(b))
        return np.float32(a.value), np.float32(b.value)

    @property
    def phase_q15(self):
        a, b = C.c_int16(), C.c_int16()
        self.lib().orc_xlating_phase_q15(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    @property
    def rtaps(self):
        T = self.lib().orc_xlating_taps_len(self.h)
        return np.ctypeslib.as_array(self.lib().orc_xlating_rtaps(self.h), shape=(2 * T,)).copy().view(np.complex64)

    @property
    def rtaps_q15(self):
        T = self.lib().orc_xlating_taps_len(self.h)
        return np.ctypeslib.as_array(self.lib().orc_xlating_rtaps_q15(self.h), shape=(T, 2)).copy()


def population(decimation, taps, center_freqs, sampling_freq, max_input, fmt, blocks, nblocks, nwarm=0, skip_fresh=0,
               skip_calls=0, sum_mode=0, threads=0):
    """oracle/population.c: len(center_freqs) independent filters over the same blocks on the host cores.
    blocks: 1-D array holding (nwarm + nblocks) equal blocks back to back (scalar elements, I,Q interleaved); the outputs
    of the last nblocks are returned as a list of complex64 arrays, one per client (the concatenation of its calls)."""
    L = Oracle.lib()
    fn = L.orc_population_cf32
    fn.argtypes = [C.c_uint32, _c_float_p, C.c_size_t, C.POINTER(C.c_int32), C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, C.c_size_t,
                   C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_int, _c_float_p, C.c_size_t, C.POINTER(C.c_size_t),
                   C.c_int]
    fn.restype = C.c_int
    npdt, _ = Oracle.IN_FMTS[fmt]
    blocks = np.ascontiguousarray(blocks, dtype=npdt)
    total = nwarm + nblocks
    assert total > 0 and blocks.size % total == 0
    block_len = blocks.size // total
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    fcs = np.ascontiguousarray(center_freqs, dtype=np.int32)
    n = fcs.size
    cap = nblocks * (block_len // 2 // decimation + 1)
    out = np.zeros((n, cap), np.complex64)
    lens = np.zeros(n, np.uintp)
    code = fn(decimation, taps.ctypes.data_as(_c_float_p), taps.size, fcs.ctypes.data_as(C.POINTER(C.c_int32)), n, sampling_freq,
              max_input, list(Oracle.IN_FMTS).index(fmt), skip_fresh, skip_calls, blocks.ctypes.data, block_len, nwarm, nblocks,
              sum_mode, out.ctypes.data_as(_c_float_p), cap, lens.ctypes.data_as(C.POINTER(C.c_size_t)), threads)
    if code != 0:
        raise RuntimeError("orc_population_cf32 failed")
    return [out[c, :int(lens[c])] for c in range(n)]


# --------------------------------------------------------------------------- reference build


class RefLib:
    """The unmodified reference (oracle/_ref/libref_{canon,fast}.so): same C API as src/xlating.h."""

    _libs = {}

    @classmethod
    def available(cls, flavour="canon"):
        return os.path.exists(os.path.join(HERE, "_ref", f"libref_{flavour}.so"))

    @classmethod
    def lib(cls, flavour="canon"):
        if flavour not in cls._libs:
            L = _load(os.path.join(HERE, "_ref", f"libref_{flavour}.so"))
            L.create_low_pass_filter.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_c_float_p), C.POINTER(C.c_size_t)]
            L.create_low_pass_filter.restype = C.c_int
            L.create_frequency_xlating_filter.argtypes = [C.c_uint32, _c_float_p, C.c_size_t, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
            L.create_frequency_xlating_filter.restype = C.c_int
            L.destroy_xlating.argtypes = [C.c_void_p]
            for var in ("native", "optimized"):
                for fmt, ct in (("cu8", C.c_uint8), ("cs8", C.c_int8), ("cs16", C.c_int16)):
                    getattr(L, f"process_{var}_{fmt}_cf32").argtypes = [C.POINTER(ct), C.c_size_t, C.POINTER(_c_float_p), C.POINTER(C.c_size_t), C.c_void_p]
                    getattr(L, f"process_{var}_{fmt}_cs16").argtypes = [C.POINTER(ct), C.c_size_t, C.POINTER(_c_i16_p), C.POINTER(C.c_size_t), C.c_void_p]
            cls._libs[flavour] = L
        return cls._libs[flavour]

    @classmethod
    def simd_status(cls, flavour="canon"):
        return C.c_char_p.in_dll(cls.lib(flavour), "SIMD_STATUS").value.decode()

    @classmethod
    def lpf(cls, gain, fs, cutoff, tw, flavour="canon"):
        L = cls.lib(flavour)
        p = _c_float_p()
        n = C.c_size_t(0)
        code = L.create_low_pass_filter(gain, fs, cutoff, tw, C.byref(p), C.byref(n))
        if code != 0:
            return code, None
        taps = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
        C.CDLL(None).free(p)
        return 0, taps

    def __init__(self, decimation, taps, center_freq, sampling_freq, max_input, flavour="canon", variant="native"):
        self.L = self.lib(flavour)
        self.variant = variant
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        # the reference takes ownership of `taps` and free()s it: hand it a malloc'd copy
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        libc.malloc.argtypes = [C.c_size_t]
        buf = libc.malloc(max(4, taps.nbytes))
        C.memmove(buf, taps.ctypes.data, taps.nbytes)
        h = C.c_void_p()
        code = self.L.create_frequency_xlating_filter(decimation, C.cast(buf, _c_float_p), taps.size, center_freq, sampling_freq, max_input, C.byref(h))
        if code != 0:
            raise RuntimeError(f"create_frequency_xlating_filter -> {code}")
        self.h = h

    def close(self):
        if self.h:
            self.L.destroy_xlating(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    _CT = {"cu8": (np.uint8, C.c_uint8), "cs8": (np.int8, C.c_int8), "cs16": (np.int16, C.c_int16)}

    def process(self, fmt, x, out="cf32", variant=None):
        npdt, ct = self._CT[fmt]
        x = np.ascontiguousarray(x, dtype=npdt)
        n = C.c_size_t(0)
        fn = getattr(self.L, f"process_{variant or self.variant}_{fmt}_{out}")
        if out == "cf32":
            p = _c_float_p()
            fn(x.ctypes.data_as(C.POINTER(ct)), x.size, C.byref(p), C.byref(n), self.h)
            if n.value == 0:
                return np.zeros(0, np.complex64)
            return np.ctypeslib.as_array(p, shape=(2 * n.value,)).copy().view(np.complex64)
        p = _c_i16_p()
        fn(x.ctypes.data_as(C.POINTER(ct)), x.size, C.byref(p), C.byref(n), self.h)
        if n.value == 0:
            return np.zeros((0, 2), np.int16)
        return np.ctypeslib.as_array(p, shape=(n.value, 2)).copy()

    def process_raw(self, fmt, x, out="cf32", variant=None):
        """Timing helper: no output copy. Returns output_len."""
        npdt, ct = self._CT[fmt]
        n = C.c_size_t(0)
        fn = getattr(self.L, f"process_{variant or self.variant}_{fmt}_{out}")
        p = _c_float_p() if out == "cf32" else _c_i16_p()
        fn(x.ctypes.data_as(C.POINTER(ct)), x.size, C.byref(p), C.byref(n), self.h)
        return n.value
