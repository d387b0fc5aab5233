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
