"""sdr-server_amd -- MI355X-native implementation of sdr-server's frequency-xlating FIR hot path.

The product is the C-ABI shared library `lib/libxlating_hip.so` (sources in csrc/, headers in ../include/):
hand-written HIP kernels for gfx950 behind the reference's own `xlating.h` / `lpf.h` API plus the batched
fan-out API of `xlating_batch.h`.  This Python package is only a thin ctypes host used by the tests and by
bench.py; it mirrors the reference interface one to one (same names, argument meaning, error behaviour):

    code, taps = create_low_pass_filter(1.0, 2016000, 24000, 9600)          # src/lpf.h:6
    f = XlatingFilter(42, taps, -12000, 2016000, 262144)                    # create_frequency_xlating_filter
    y = f.process("native", "cu8", "cf32", block)                           # process_native_cu8_cf32
    f.close()                                                               # destroy_xlating

There is NO CPU arithmetic path: if the library or a HIP device is missing, construction raises.
(The directory name contains '-'; import it through the top-level alias module `sdr_server_amd`.)
"""
import ctypes as C
import os

import numpy as np

from .build import build_library, library_path


class WireRequest(C.Structure):
    """include/xlating_wire.h xlating_wire_request"""
    _fields_ = [("center_freq", C.c_uint32), ("sampling_rate", C.c_uint32), ("band_freq", C.c_uint32), ("destination", C.c_uint8)]


class WireAdmission(C.Structure):
    """include/xlating_wire.h xlating_wire_admission"""
    _fields_ = [("decimation", C.c_uint32), ("center_offset", C.c_int32), ("lpf_cutoff", C.c_uint32), ("lpf_transition", C.c_uint32)]

_c_float_p = C.POINTER(C.c_float)
_c_i16_p = C.POINTER(C.c_int16)

FMT = {"cu8": 0, "cs8": 1, "cs16": 2, "cf32": 3}
MODE = {"native": 0, "optimized": 1, "q15": 2, "optimized_x86": 3, "optimized_x86_fma": 4}
_NP = {"cu8": np.uint8, "cs8": np.int8, "cs16": np.int16, "cf32": np.float32}
_CT = {"cu8": C.c_uint8, "cs8": C.c_int8, "cs16": C.c_int16, "cf32": C.c_float}

# every symbol include/xlating.h, include/lpf.h and include/xlating_batch.h declare
EXPORTED_SYMBOLS = (
    ["SIMD_STATUS", "create_frequency_xlating_filter", "destroy_xlating", "create_low_pass_filter"]
    + [f"process_{v}_{i}_{o}" for v in ("native", "optimized") for i in ("cu8", "cs8", "cs16") for o in ("cf32", "cs16")]
    + ["process_native_cf32_cf32", "process_optimized_cf32_cf32", "xlating_set_optimized_x86"]
    + ["xlating_batch_create", "xlating_batch_create_grouped", "xlating_batch_set_option", "xlating_batch_process_host_group",
       "xlating_batch_process_device_group", "xlating_batch_process_device_group_ev", "xlating_batch_output_len_block", "xlating_batch_add_client", "xlating_batch_remove_client", "xlating_batch_num_clients",
       "xlating_batch_process_host", "xlating_batch_process_device", "xlating_batch_output_len", "xlating_batch_fetch",
       "xlating_batch_output_host", "xlating_batch_output_host_cs16", "xlating_batch_output_device", "xlating_batch_client_phase", "xlating_batch_sync",
       "xlating_batch_query", "xlating_batch_record_event", "xlating_batch_timing", "xlating_batch_timing_read", "xlating_batch_timing_polyphase", "xlating_batch_timing_stride", "xlating_batch_describe", "xlating_batch_destroy",
       "xlating_hip_device_info"]
    + ["xlating_sinks_create", "xlating_sinks_attach_fd", "xlating_sinks_attach_file", "xlating_sinks_write",
       "xlating_sinks_submit", "xlating_sinks_failed", "xlating_sinks_flush", "xlating_sinks_detach", "xlating_sinks_stats",
       "xlating_sinks_destroy"]
    + ["xlating_wire_parse_header", "xlating_wire_parse_request", "xlating_wire_build_request", "xlating_wire_build_response",
       "xlating_wire_build_header", "xlating_wire_parse_response", "xlating_wire_admit", "xlating_wire_add_client"]
)

MULTI_SYMBOLS = ["xlating_multi_unique_id", "xlating_multi_create_rank", "xlating_multi_create_local", "xlating_multi_world",
                 "xlating_multi_local", "xlating_multi_add_client", "xlating_multi_engine", "xlating_multi_feed",
                 "xlating_multi_feed_done", "xlating_multi_feed_query", "xlating_multi_feed_wait_on_stream",
                 "xlating_multi_feed_timing", "xlating_multi_feed_timing_read", "xlating_multi_comm_count",
                 "xlating_multi_sync", "xlating_multi_destroy"]

_lib = None
_mlib = None


def multi_library_path():
    return os.path.join(os.path.dirname(library_path()), "libxlating_multi.so")


def multi_lib():
    """include/xlating_multi.h: the C host of the multi-GPU path (engines + RCCL broadcast).  Loads librccl."""
    global _mlib
    if _mlib is not None:
        return _mlib
    lib()
    path = multi_library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    M = C.CDLL(path)
    M.xlating_multi_unique_id.argtypes = [C.c_void_p]
    M.xlating_multi_unique_id.restype = C.c_int
    M.xlating_multi_create_rank.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint, C.c_int,
                                            C.POINTER(C.c_void_p)]
    M.xlating_multi_create_rank.restype = C.c_int
    M.xlating_multi_create_local.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_uint32, C.c_int, C.c_uint32, C.c_uint,
                                             C.POINTER(C.c_void_p)]
    M.xlating_multi_create_local.restype = C.c_int
    M.xlating_multi_world.argtypes = [C.c_void_p]
    M.xlating_multi_world.restype = C.c_int
    M.xlating_multi_local.argtypes = [C.c_void_p]
    M.xlating_multi_local.restype = C.c_int
    M.xlating_multi_add_client.argtypes = [C.c_void_p, C.c_int, C.c_uint32, _c_float_p, C.c_size_t, C.c_int32]
    M.xlating_multi_add_client.restype = C.c_int
    M.xlating_multi_engine.argtypes = [C.c_void_p, C.c_int]
    M.xlating_multi_engine.restype = C.c_void_p
    M.xlating_multi_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_int]
    M.xlating_multi_feed.restype = C.c_int
    M.xlating_multi_sync.argtypes = [C.c_void_p]
    M.xlating_multi_sync.restype = C.c_int
    for name in ("xlating_multi_feed_done", "xlating_multi_feed_query"):
        getattr(M, name).argtypes = [C.c_void_p]
        getattr(M, name).restype = C.c_int
    M.xlating_multi_feed_wait_on_stream.argtypes = [C.c_void_p, C.c_void_p]
    M.xlating_multi_feed_wait_on_stream.restype = C.c_int
    M.xlating_multi_feed_timing.argtypes = [C.c_void_p, C.c_int]
    M.xlating_multi_feed_timing.restype = C.c_int
    M.xlating_multi_feed_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    M.xlating_multi_feed_timing_read.restype = C.c_int
    M.xlating_multi_comm_count.argtypes = [C.c_void_p]
    M.xlating_multi_comm_count.restype = C.c_int
    M.xlating_multi_destroy.argtypes = [C.c_void_p]
    M.xlating_multi_destroy.restype = None
    _mlib = M
    return M


def lib():
    """Load (never build) the in-tree shared library and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    # PyTorch bundles its own copy of the HIP runtime.  If this library (linked against /opt/rocm's) initialises HIP first
    # and torch is imported later in the same process, torch ends up on a second runtime that sees no GPU ("No HIP GPUs are
    # available"); with torch's copy loaded first both share it.  Tests and bench.py use torch for device buffers, so:
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional for this host; a pure C deployment never has this problem
        pass
    L = C.CDLL(path)
    L.create_low_pass_filter.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_c_float_p), C.POINTER(C.c_size_t)]
    L.create_low_pass_filter.restype = C.c_int
    L.create_frequency_xlating_filter.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.create_frequency_xlating_filter.restype = C.c_int
    L.destroy_xlating.argtypes = [C.c_void_p]
    L.destroy_xlating.restype = None
    L.xlating_set_optimized_x86.argtypes = [C.c_void_p, C.c_int]
    L.xlating_set_optimized_x86.restype = C.c_int
    for v in ("native", "optimized"):
        for i in ("cu8", "cs8", "cs16", "cf32"):
            fn = getattr(L, f"process_{v}_{i}_cf32")
            fn.argtypes = [C.POINTER(_CT[i]), C.c_size_t, C.POINTER(_c_float_p), C.POINTER(C.c_size_t), C.c_void_p]
            fn.restype = None
            if i != "cf32":
                fn = getattr(L, f"process_{v}_{i}_cs16")
                fn.argtypes = [C.POINTER(_CT[i]), C.c_size_t, C.POINTER(_c_i16_p), C.POINTER(C.c_size_t), C.c_void_p]
                fn.restype = None
    L.xlating_batch_create.argtypes = [C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
    L.xlating_batch_create.restype = C.c_int
    L.xlating_batch_create_grouped.argtypes = [C.c_uint32, C.c_int, C.c_uint32, C.c_uint, C.c_int, C.POINTER(C.c_void_p)]
    L.xlating_batch_create_grouped.restype = C.c_int
    L.xlating_batch_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
    L.xlating_batch_set_option.restype = C.c_int
    L.xlating_batch_process_host_group.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_int]
    L.xlating_batch_process_host_group.restype = C.c_int
    L.xlating_batch_process_device_group.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_int, C.c_void_p]
    L.xlating_batch_process_device_group.restype = C.c_int
    L.xlating_batch_output_len_block.argtypes = [C.c_void_p, C.c_int, C.c_uint]
    L.xlating_batch_output_len_block.restype = C.c_size_t
    L.xlating_batch_add_client.argtypes = [C.c_void_p, C.c_uint32, _c_float_p, C.c_size_t, C.c_int32]
    L.xlating_batch_add_client.restype = C.c_int
    L.xlating_batch_remove_client.argtypes = [C.c_void_p, C.c_int]
    L.xlating_batch_remove_client.restype = C.c_int
    L.xlating_batch_num_clients.argtypes = [C.c_void_p]
    L.xlating_batch_num_clients.restype = C.c_int
    L.xlating_batch_process_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.xlating_batch_process_host.restype = C.c_int
    L.xlating_batch_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    L.xlating_batch_process_device.restype = C.c_int
    L.xlating_batch_output_len.argtypes = [C.c_void_p, C.c_int]
    L.xlating_batch_output_len.restype = C.c_size_t
    L.xlating_batch_fetch.argtypes = [C.c_void_p]
    L.xlating_batch_fetch.restype = C.c_int
    L.xlating_batch_output_host.argtypes = [C.c_void_p, C.c_int, C.POINTER(_c_float_p), C.POINTER(C.c_size_t)]
    L.xlating_batch_output_host.restype = C.c_int
    L.xlating_batch_output_host_cs16.argtypes = [C.c_void_p, C.c_int, C.POINTER(_c_i16_p), C.POINTER(C.c_size_t)]
    L.xlating_batch_output_host_cs16.restype = C.c_int
    L.xlating_batch_output_device.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.xlating_batch_output_device.restype = C.c_int
    L.xlating_batch_client_phase.argtypes = [C.c_void_p, C.c_int, _c_float_p, _c_float_p]
    L.xlating_batch_client_phase.restype = C.c_int
    L.xlating_batch_sync.argtypes = [C.c_void_p]
    L.xlating_batch_sync.restype = C.c_int
    L.xlating_batch_timing.argtypes = [C.c_void_p, C.c_int]
    L.xlating_batch_timing.restype = C.c_int
    L.xlating_batch_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    L.xlating_batch_timing_read.restype = C.c_int
    L.xlating_sinks_create.argtypes = [C.c_uint, C.c_size_t, C.POINTER(C.c_void_p)]
    L.xlating_sinks_create.restype = C.c_int
    L.xlating_sinks_attach_fd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.xlating_sinks_attach_fd.restype = C.c_int
    L.xlating_sinks_attach_file.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    L.xlating_sinks_attach_file.restype = C.c_int
    L.xlating_sinks_write.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.xlating_sinks_write.restype = C.c_int
    L.xlating_sinks_submit.argtypes = [C.c_void_p, C.c_void_p]
    L.xlating_sinks_submit.restype = C.c_int
    L.xlating_sinks_failed.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_size_t]
    L.xlating_sinks_failed.restype = C.c_size_t
    L.xlating_sinks_flush.argtypes = [C.c_void_p]
    L.xlating_sinks_flush.restype = C.c_int
    L.xlating_sinks_detach.argtypes = [C.c_void_p, C.c_int]
    L.xlating_sinks_detach.restype = C.c_int
    L.xlating_sinks_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.xlating_sinks_stats.restype = None
    L.xlating_sinks_destroy.argtypes = [C.c_void_p]
    L.xlating_sinks_destroy.restype = None
    L.xlating_wire_parse_header.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint8)]
    L.xlating_wire_parse_header.restype = C.c_int
    L.xlating_wire_parse_request.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(WireRequest)]
    L.xlating_wire_parse_request.restype = C.c_int
    L.xlating_wire_build_request.argtypes = [C.POINTER(WireRequest), C.c_char_p]
    L.xlating_wire_build_request.restype = C.c_size_t
    L.xlating_wire_build_response.argtypes = [C.c_uint8, C.c_uint32, C.c_char_p]
    L.xlating_wire_build_response.restype = C.c_size_t
    L.xlating_wire_build_header.argtypes = [C.c_uint8, C.c_char_p]
    L.xlating_wire_build_header.restype = C.c_size_t
    L.xlating_wire_parse_response.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)]
    L.xlating_wire_parse_response.restype = C.c_int
    L.xlating_wire_admit.argtypes = [C.POINTER(WireRequest), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(WireAdmission),
                                     C.POINTER(C.c_uint32)]
    L.xlating_wire_admit.restype = C.c_int
    L.xlating_wire_add_client.argtypes = [C.c_void_p, C.POINTER(WireAdmission), C.c_uint32]
    L.xlating_wire_add_client.restype = C.c_int
    L.xlating_batch_timing_stride.argtypes = [C.c_void_p, C.c_uint]
    L.xlating_batch_timing_stride.restype = C.c_int
    L.xlating_batch_timing_polyphase.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    L.xlating_batch_timing_polyphase.restype = C.c_int
    L.xlating_batch_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.xlating_batch_describe.restype = C.c_int
    L.xlating_batch_destroy.argtypes = [C.c_void_p]
    L.xlating_batch_destroy.restype = None
    L.xlating_hip_device_info.argtypes = []
    L.xlating_hip_device_info.restype = C.c_char_p
    _lib = L
    return L


_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]
_libc.free.argtypes = [C.c_void_p]


def simd_status():
    """The exported SIMD_STATUS string (reference src/xlating.c:145-268; printed by src/main.c:23)."""
    return C.c_char_p.in_dll(lib(), "SIMD_STATUS").value.decode()


def device_info():
    return lib().xlating_hip_device_info().decode()


def create_low_pass_filter(gain, sampling_freq, cutoff_freq, transition_width):
    """reference src/lpf.h:6 -> (code, float32 taps | None)"""
    p = _c_float_p()
    n = C.c_size_t(0)
    code = lib().create_low_pass_filter(gain, sampling_freq, cutoff_freq, transition_width, C.byref(p), C.byref(n))
    if code != 0:
        return code, None
    taps = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    _libc.free(p)
    return 0, taps


class XlatingError(RuntimeError):
    def __init__(self, what, code):
        super().__init__(f"{what} failed with code {code}")
        self.code = code


class XlatingFilter:
    """One `xlating *` handle (reference src/xlating.h:8-38)."""

    def __init__(self, decimation, taps, center_freq, sampling_freq, max_input_buffer_length):
        L = lib()
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        # create() takes ownership of a malloc'd taps array (xlating.c:508, freed at :600-602)
        buf = _libc.malloc(max(4, taps.nbytes))
        C.memmove(buf, taps.ctypes.data, taps.nbytes)
        h = C.c_void_p()
        code = L.create_frequency_xlating_filter(decimation, buf, taps.size, center_freq, sampling_freq,
                                                 max_input_buffer_length, C.byref(h))
        if code != 0:
            if code == -1:  # taps_len == 0: taps not consumed (xlating.c:496-498)
                _libc.free(buf)
            raise XlatingError("create_frequency_xlating_filter", code)
        self.h = h
        self.D = decimation

    def process(self, variant, in_fmt, out_fmt, x):
        """process_<variant>_<in_fmt>_<out_fmt>(x) -> complex64[K] (cf32) or int16[K, 2] (cs16).
        x: 1-D array of scalar elements (I,Q interleaved); len(x) is the C API's input_len."""
        x = np.ascontiguousarray(x, dtype=_NP[in_fmt])
        fn = getattr(lib(), f"process_{variant}_{in_fmt}_{out_fmt}")
        n = C.c_size_t(0)
        if out_fmt == "cf32":
            p = _c_float_p()
            fn(x.ctypes.data_as(C.POINTER(_CT[in_fmt])), x.size, C.byref(p), C.byref(n), self.h)
            if n.value == 0:
                return np.zeros(0, np.complex64)
            return np.ctypeslib.as_array(p, shape=(2 * n.value,)).copy().view(np.complex64)
        p = _c_i16_p()
        fn(x.ctypes.data_as(C.POINTER(_CT[in_fmt])), x.size, C.byref(p), C.byref(n), self.h)
        if n.value == 0:
            return np.zeros((0, 2), np.int16)
        return np.ctypeslib.as_array(p, shape=(n.value, 2)).copy()

    def set_optimized_x86(self, on=1):
        """process_optimized_* follows the reference's x86 AVX build: the phase is never renormalised (xlating.c:338-339);
        on = 2: the build compiled with FMA (its contracted phase step)."""
        code = lib().xlating_set_optimized_x86(self.h, int(on))
        if code != 0:
            raise XlatingError("xlating_set_optimized_x86", code)

    def close(self):
        if getattr(self, "h", None):
            lib().destroy_xlating(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchEngine:
    """`xlating_batch *` (include/xlating_batch.h): one IQ stream, many clients, one GPU."""

    def __init__(self, sampling_freq, in_fmt, max_input_buffer_length, device=-1, group_blocks=1):
        h = C.c_void_p()
        code = lib().xlating_batch_create_grouped(sampling_freq, FMT[in_fmt], max_input_buffer_length, group_blocks, device,
                                                  C.byref(h))
        if code != 0:
            raise XlatingError("xlating_batch_create", code)
        self.h = h
        self.in_fmt = in_fmt

    def set_option(self, name, value):
        code = lib().xlating_batch_set_option(self.h, name.encode(), int(value))
        if code != 0:
            raise XlatingError(f"xlating_batch_set_option({name})", code)

    def process_host_group(self, x, nblocks, variant="native"):
        """x holds nblocks equal blocks back to back; results == nblocks successive process_host calls."""
        x = np.ascontiguousarray(x, dtype=_NP[self.in_fmt])
        assert x.size % nblocks == 0
        code = lib().xlating_batch_process_host_group(self.h, x.ctypes.data, x.size // nblocks, nblocks, MODE[variant])
        if code != 0:
            raise XlatingError("xlating_batch_process_host_group", code)

    def process_device_group(self, d_ptr, input_len, nblocks, variant="native", stream=0):
        """stream: a hipStream_t value, 0 (HIP's default stream) or "engine" (XL_STREAM_ENGINE: the engine's own compute
        stream, CU-masked for calls whose NCO chain runs on the side stream; wait with sync())."""
        sp = C.c_void_p(-1) if stream == "engine" else (C.c_void_p(stream) if stream else None)
        code = lib().xlating_batch_process_device_group(self.h, C.c_void_p(d_ptr), input_len, nblocks, MODE[variant], sp)
        if code != 0:
            raise XlatingError("xlating_batch_process_device_group", code)

    def output_len_block(self, cid, block):
        return lib().xlating_batch_output_len_block(self.h, cid, block)

    def add_client(self, decimation, taps, center_freq):
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        cid = lib().xlating_batch_add_client(self.h, decimation, taps.ctypes.data_as(_c_float_p), taps.size, center_freq)
        if cid < 0:
            raise XlatingError("xlating_batch_add_client", cid)
        return cid

    def remove_client(self, cid):
        code = lib().xlating_batch_remove_client(self.h, cid)
        if code != 0:
            raise XlatingError("xlating_batch_remove_client", code)

    @property
    def num_clients(self):
        return lib().xlating_batch_num_clients(self.h)

    def process_host(self, x, variant="native"):
        x = np.ascontiguousarray(x, dtype=_NP[self.in_fmt])
        code = lib().xlating_batch_process_host(self.h, x.ctypes.data, x.size, MODE[variant])
        if code != 0:
            raise XlatingError("xlating_batch_process_host", code)

    def process_device(self, d_ptr, input_len, variant="native", stream=0):
        code = lib().xlating_batch_process_device(self.h, C.c_void_p(d_ptr), input_len, MODE[variant],
                                                  C.c_void_p(stream) if stream else None)
        if code != 0:
            raise XlatingError("xlating_batch_process_device", code)

    def fetch(self):
        code = lib().xlating_batch_fetch(self.h)
        if code != 0:
            raise XlatingError("xlating_batch_fetch", code)

    def output(self, cid):
        p = _c_float_p()
        n = C.c_size_t(0)
        code = lib().xlating_batch_output_host(self.h, cid, C.byref(p), C.byref(n))
        if code != 0:
            raise XlatingError("xlating_batch_output_host", code)
        if n.value == 0:
            return np.zeros(0, np.complex64)
        return np.ctypeslib.as_array(p, shape=(2 * n.value,)).copy().view(np.complex64)

    def output_cs16(self, cid):
        """After a "q15" call + fetch(): int16[K, 2]."""
        p = _c_i16_p()
        n = C.c_size_t(0)
        code = lib().xlating_batch_output_host_cs16(self.h, cid, C.byref(p), C.byref(n))
        if code != 0:
            raise XlatingError("xlating_batch_output_host_cs16", code)
        if n.value == 0:
            return np.zeros((0, 2), np.int16)
        return np.ctypeslib.as_array(p, shape=(n.value, 2)).copy()

    def output_len(self, cid):
        return lib().xlating_batch_output_len(self.h, cid)

    def output_device(self, cid):
        p = C.c_void_p()
        n = C.c_size_t(0)
        code = lib().xlating_batch_output_device(self.h, cid, C.byref(p), C.byref(n))
        if code != 0:
            raise XlatingError("xlating_batch_output_device", code)
        return p.value, n.value

    def phase(self, cid):
        a, b = C.c_float(), C.c_float()
        code = lib().xlating_batch_client_phase(self.h, cid, C.byref(a), C.byref(b))
        if code != 0:
            raise XlatingError("xlating_batch_client_phase", code)
        return np.float32(a.value), np.float32(b.value)

    def sync(self):
        code = lib().xlating_batch_sync(self.h)
        if code != 0:
            raise XlatingError("xlating_batch_sync", code)

    def timing(self, enable):
        """False/0 off, True/1 bracket every block's launches, 2 also time the three polyphase launches separately."""
        lib().xlating_batch_timing(self.h, int(enable))

    def timing_stride(self, every_n):
        """Bracket only every n-th block with events (an event pair costs a few us of stream time)."""
        lib().xlating_batch_timing_stride(self.h, int(every_n))

    def timing_polyphase(self, reset=True):
        """-> (n_timed_blocks, [forward_ms_total, mix_ms_total, inverse_ms_total]) -- needs timing(2)"""
        a = (C.c_double * 3)()
        n = lib().xlating_batch_timing_polyphase(self.h, a, 1 if reset else 0)
        if n < 0:
            raise XlatingError("xlating_batch_timing_polyphase", n)
        return n, [a[0], a[1], a[2]]

    def describe(self):
        """The resident plan in one line (which classes take which arithmetic path)."""
        buf = C.create_string_buffer(1024)
        n = lib().xlating_batch_describe(self.h, buf, 1024)
        if n < 0:
            raise XlatingError("xlating_batch_describe", n)
        return buf.value.decode()

    def timing_read(self, reset=True):
        """-> (n_timed_blocks, fir_ms_total, nco_ms_total)"""
        a, b = C.c_double(0), C.c_double(0)
        n = lib().xlating_batch_timing_read(self.h, C.byref(a), C.byref(b), 1 if reset else 0)
        if n < 0:
            raise XlatingError("xlating_batch_timing_read", n)
        return n, a.value, b.value

    def close(self):
        if getattr(self, "h", None):
            lib().xlating_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiHost:
    """`xlating_multi *` (include/xlating_multi.h): the C host of the multi-GPU path.  One process per GPU
    (rank / world / 128-byte id from `MultiHost.unique_id()` on rank 0) or one process driving `ngpus` GPUs."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        code = multi_lib().xlating_multi_unique_id(buf)
        if code != 0:
            raise XlatingError("xlating_multi_unique_id", code)
        return buf.raw

    def __init__(self, sampling_freq, in_fmt, max_input_buffer_length, group_blocks=1, rank=0, world=1, uid=None, device=-1,
                 ngpus=None):
        h = C.c_void_p()
        M = multi_lib()
        if ngpus is not None:
            code = M.xlating_multi_create_local(ngpus, None, sampling_freq, FMT[in_fmt], max_input_buffer_length, group_blocks,
                                                C.byref(h))
        else:
            idbuf = C.create_string_buffer(uid, 128) if uid is not None else None
            code = M.xlating_multi_create_rank(rank, world, idbuf, sampling_freq, FMT[in_fmt], max_input_buffer_length,
                                               group_blocks, device, C.byref(h))
        if code != 0:
            raise XlatingError("xlating_multi_create", code)
        self.h = h
        self.in_fmt = in_fmt
        self.world = M.xlating_multi_world(h)

    def engine(self, gpu):
        """The BatchEngine of job GPU `gpu` if this process drives it (owned by the host: do not close it)."""
        p = multi_lib().xlating_multi_engine(self.h, gpu)
        if not p:
            return None
        e = BatchEngine.__new__(BatchEngine)
        e.h = C.c_void_p(p)
        e.in_fmt = self.in_fmt
        e.close = lambda: None
        return e

    def add_client(self, global_client, decimation, taps, center_freq):
        """-> engine-local client id, or None when another process drives the client's GPU (global_client mod world)."""
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        cid = multi_lib().xlating_multi_add_client(self.h, global_client, decimation, taps.ctypes.data_as(_c_float_p), taps.size,
                                                   center_freq)
        if cid == -2:
            return None
        if cid < 0:
            raise XlatingError("xlating_multi_add_client", cid)
        return cid

    def feed(self, d_src, input_len, nblocks, variant="optimized"):
        code = multi_lib().xlating_multi_feed(self.h, C.c_void_p(d_src) if d_src else None, input_len, nblocks, MODE[variant])
        if code != 0:
            raise XlatingError("xlating_multi_feed", code)

    def feed_done(self):
        """Block until the source buffer of the latest feed may be overwritten (xlating_multi_feed_done)."""
        code = multi_lib().xlating_multi_feed_done(self.h)
        if code != 0:
            raise XlatingError("xlating_multi_feed_done", code)

    def feed_query(self):
        code = multi_lib().xlating_multi_feed_query(self.h)
        if code < 0:
            raise XlatingError("xlating_multi_feed_query", code)
        return bool(code)

    def feed_wait_on_stream(self, stream=0):
        code = multi_lib().xlating_multi_feed_wait_on_stream(self.h, C.c_void_p(stream) if stream else None)
        if code != 0:
            raise XlatingError("xlating_multi_feed_wait_on_stream", code)

    def feed_timing(self, enable=True):
        """Bracket every broadcast with HIP events on the communication stream (xlating_multi_feed_timing)."""
        code = multi_lib().xlating_multi_feed_timing(self.h, 1 if enable else 0)
        if code != 0:
            raise XlatingError("xlating_multi_feed_timing", code)

    def feed_timing_read(self, reset=True):
        """-> (feeds measured, broadcast ms in total, of which hidden behind the previous feed's filtering)."""
        b, hd = C.c_double(0.0), C.c_double(0.0)
        n = multi_lib().xlating_multi_feed_timing_read(self.h, C.byref(b), C.byref(hd), 1 if reset else 0)
        if n < 0:
            raise XlatingError("xlating_multi_feed_timing_read", n)
        return n, b.value, hd.value

    def comm_count(self):
        """Ranks of the RCCL communicator the host broadcasts over (0: none)."""
        n = multi_lib().xlating_multi_comm_count(self.h)
        if n < 0:
            raise XlatingError("xlating_multi_comm_count", n)
        return n

    def sync(self):
        code = multi_lib().xlating_multi_sync(self.h)
        if code != 0:
            raise XlatingError("xlating_multi_sync", code)

    def close(self):
        if getattr(self, "h", None):
            multi_lib().xlating_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Sinks:
    """ctypes mirror of include/xlating_sinks.h (per-client output delivery of the batched path; host-only)."""

    def __init__(self, writer_threads=4, queue_bytes=16 * 25600):
        h = C.c_void_p()
        code = lib().xlating_sinks_create(writer_threads, queue_bytes, C.byref(h))
        if code != 0:
            raise XlatingError("xlating_sinks_create", code)
        self.h = h

    def attach_fd(self, client_id, fd, close_on_detach=False):
        return lib().xlating_sinks_attach_fd(self.h, client_id, fd, 1 if close_on_detach else 0)

    def attach_file(self, client_id, base_path, use_gzip=False):
        return lib().xlating_sinks_attach_file(self.h, client_id, str(base_path).encode(), 1 if use_gzip else 0)

    def write(self, client_id, samples):
        a = np.ascontiguousarray(samples, dtype=np.complex64)
        return lib().xlating_sinks_write(self.h, client_id, a.ctypes.data_as(C.c_void_p), a.size)

    def submit(self, engine):
        return lib().xlating_sinks_submit(self.h, engine.h)

    def failed(self, cap=4096):
        ids = (C.c_int * cap)()
        n = lib().xlating_sinks_failed(self.h, ids, cap)
        return [ids[i] for i in range(n)]

    def flush(self):
        return lib().xlating_sinks_flush(self.h)

    def detach(self, client_id):
        return lib().xlating_sinks_detach(self.h, client_id)

    def stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        lib().xlating_sinks_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def close(self):
        if getattr(self, "h", None):
            lib().xlating_sinks_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- include/xlating_wire.h (client wire format + admission rules; host-only)
def wire_build_request(center_freq, sampling_rate, band_freq, destination):
    req = WireRequest(center_freq & 0xFFFFFFFF, sampling_rate, band_freq, destination)
    buf = C.create_string_buffer(15)
    n = lib().xlating_wire_build_request(C.byref(req), buf)
    return buf.raw[:n]


def wire_build_response(status, details):
    buf = C.create_string_buffer(7)
    n = lib().xlating_wire_build_response(status, details, buf)
    return buf.raw[:n]


def wire_build_header(mtype):
    buf = C.create_string_buffer(2)
    n = lib().xlating_wire_build_header(mtype, buf)
    return buf.raw[:n]


def wire_parse_header(data):
    t = C.c_uint8(0)
    return lib().xlating_wire_parse_header(bytes(data), len(data), C.byref(t)), t.value


def wire_parse_request(data):
    req = WireRequest()
    code = lib().xlating_wire_parse_request(bytes(data), len(data), C.byref(req))
    return code, req


def wire_parse_response(data):
    st, det = C.c_uint8(0), C.c_uint32(0)
    code = lib().xlating_wire_parse_response(bytes(data), len(data), C.byref(st), C.byref(det))
    return code, st.value, det.value


def wire_admit(req, band_sampling_rate, current_band_freq=0, lpf_cutoff_rate=5):
    """-> (code, WireAdmission, failure_details)"""
    adm, why = WireAdmission(), C.c_uint32(0)
    code = lib().xlating_wire_admit(C.byref(req), band_sampling_rate, current_band_freq, lpf_cutoff_rate, C.byref(adm), C.byref(why))
    return code, adm, why.value


def wire_add_client(engine, adm, band_sampling_rate):
    return lib().xlating_wire_add_client(engine.h, C.byref(adm), band_sampling_rate)
