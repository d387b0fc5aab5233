"""CPU test (-m "not gpu"): every plan option xlating_batch_set_option() accepts is documented in include/xlating_batch.h, and every
option the header documents exists in the engine (csrc/xl_batch.cpp) -- the header is the only place a caller learns the names."""
import os
import re

from conftest import ROOT


def test_every_engine_option_is_documented_and_vice_versa():
    hdr = open(os.path.join(ROOT, "include", "xlating_batch.h")).read()
    src = open(os.path.join(ROOT, "sdr-server_amd", "csrc", "xl_batch.cpp")).read()
    body = src[src.index('extern "C" int xlating_batch_set_option'):]
    body = body[:body.index("\n}\n")]
    accepted = set(re.findall(r'n == "([a-z_0-9]+)"', body))
    doc = hdr[hdr.index("/* Plan options"):]
    doc = doc[:doc.index("int xlating_batch_set_option")]
    documented = set(re.findall(r'"([a-z_0-9]+)"', doc))
    assert accepted, "no options found in xl_batch.cpp"
    assert accepted - documented == set(), f"accepted but not in the header: {sorted(accepted - documented)}"
    assert documented - accepted == set(), f"in the header but not accepted: {sorted(documented - accepted)}"


def test_tuning_variables_are_ignored_without_xl_testing():
    """XL_EXP_* (tuning knobs, plan forcing) reach the library only next to XL_TESTING=1 (tests and tools set it) or in -DXL_TUNING builds:
    a plain process ignores them -- its plan must not depend on stray environment; the reference's only switch is the config file's
    cpu_optimization (src/config.c:252-264) -- and says so once on stderr ("<4>" line).  No GPU needed: the gate itself is asked."""
    import subprocess
    import sys

    lib = os.path.join(ROOT, "sdr-server_amd", "lib", "libxlating_hip.so")
    code = ("import ctypes; L = ctypes.CDLL(%r); f = L.xl_exp_getenv; f.restype = ctypes.c_char_p; "
            "print(f(b'XL_EXP_POLY'), f(b'XL_EXP_MIX'), f(b'XL_EXP_NOMASK'))" % lib)
    base = {k: v for k, v in os.environ.items() if not k.startswith("XL_")}
    knobs = {"XL_EXP_POLY": "1", "XL_EXP_MIX": "3", "XL_EXP_NOMASK": "1"}
    r = subprocess.run([sys.executable, "-c", code], env=dict(base, **knobs), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.split() == ["None", "None", "None"], (r.stdout, r.stderr[-500:])
    assert r.stderr.count("<4>xlating-hip: XL_EXP_* tuning variables are set but ignored") == 1, r.stderr[-500:]
    r = subprocess.run([sys.executable, "-c", code], env=dict(base, XL_TESTING="1", **knobs), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.split() == ["b'1'", "b'3'", "b'1'"] and "<4>" not in r.stderr, (r.stdout, r.stderr[-500:])
    r = subprocess.run([sys.executable, "-c", code], env=base, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "<4>" not in r.stderr  # (nothing set: nothing to say)
    # every getenv of a tuning name in the sources goes through the gate
    import re
    for f in ("xl_batch.cpp", "xl_filter.cpp", "xl_multi.cpp", "xl_sinks.cpp"):
        src = open(os.path.join(ROOT, "sdr-server_amd", "csrc", f)).read()
        assert not re.findall(r'(?<!xl_exp_)getenv\("XL_EXP_', src), f
