"""Deterministic input generators shared by tests, fixtures and bench.

They restate the reference's test input formulas (test/utils.c:137-174, test/perf_xlating.c:36-39)
plus a seeded xorshift64* byte stream for "random" IQ.
"""
import numpy as np

XS_SEED = 0x5DEECE66D


def ramp_u8(off, n):
    """test/utils.c:137-145: in[i] = (uint8_t)(off + i)"""
    return ((off + np.arange(n, dtype=np.int64)) & 0xFF).astype(np.uint8)


def ramp_s8(off, n):
    """test/utils.c:147-155: in[i] = (int8_t)(off + i)"""
    return ((off + np.arange(n, dtype=np.int64)) & 0xFF).astype(np.uint8).view(np.int8)


def ramp_s16(off, n):
    """test/utils.c:157-165: in[i] = (int16_t)(off + i) - (int16_t)(n / 2)"""
    a = ((off + np.arange(n, dtype=np.int64)) & 0xFFFF).astype(np.uint16).view(np.int16).astype(np.int32)
    return ((a - np.int32(np.int16(n // 2))) & 0xFFFF).astype(np.uint16).view(np.int16)


def sin_f32(off, n):
    """test/utils.c:167-174: in[i] = sinf((float)(off + i)) (numpy float32 sin; used for cf32-input cases
    only after quantising to int16/32768 so that every consumer sees identical values)."""
    return np.sin((off + np.arange(n)).astype(np.float32)).astype(np.float32)


def staircase_u8(n):
    """test/perf_xlating.c:36-39 as evaluated on x86-64 gcc: in[i] = (i >> 7) & 0xFF"""
    return ((np.arange(n, dtype=np.int64) >> 7) & 0xFF).astype(np.uint8)


def xorshift_bytes(seed, n):
    """n bytes from xorshift64* (state -> x ^= x>>12; x ^= x<<25; x ^= x>>27; out = x * 0x2545F4914F6CDD1D),
    little-endian bytes of each 64-bit output.  Vectorised by running 64 interleaved lanes whose seeds are
    derived from `seed` with splitmix64, so 256 KiB blocks cost milliseconds."""
    lanes = 64
    m64 = np.uint64(0xFFFFFFFFFFFFFFFF)
    # splitmix64 seeding
    st = np.empty(lanes, dtype=np.uint64)
    z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        for i in range(lanes):
            z = (z + np.uint64(0x9E3779B97F4A7C15)) & m64
            y = z
            y = ((y ^ (y >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & m64
            y = ((y ^ (y >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & m64
            y = y ^ (y >> np.uint64(31))
            st[i] = y if y != 0 else np.uint64(1)
        words = (n + 7) // 8
        rounds = (words + lanes - 1) // lanes
        out = np.empty((rounds, lanes), dtype=np.uint64)
        x = st
        for r in range(rounds):
            x = x ^ (x >> np.uint64(12))
            x = x ^ (x << np.uint64(25))
            x = x ^ (x >> np.uint64(27))
            out[r] = x * np.uint64(0x2545F4914F6CDD1D)
    return out.reshape(-1).view(np.uint8)[:n].copy()


def xs_u8(seed, n):
    return xorshift_bytes(seed, n)


def xs_s8(seed, n):
    return xorshift_bytes(seed, n).view(np.int8)


def xs_s16(seed, n):
    return xorshift_bytes(seed, 2 * n).view(np.int16)


def hamming_sinc(ntaps, cutoff_norm):
    """Explicit float32 Hamming-windowed sinc prototype for the cf32-input extension (config 5);
    cutoff_norm = cutoff / fs.  Plain float64 math rounded once -- these taps are INPUT DATA to both the
    oracle and the HIP path (they are not claimed to equal lpf.c output)."""
    n = np.arange(ntaps, dtype=np.float64) - (ntaps - 1) / 2.0
    h = 2 * cutoff_norm * np.sinc(2 * cutoff_norm * n)
    w = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(ntaps) / (ntaps - 1))
    h = h * w
    return (h / h.sum()).astype(np.float32)
