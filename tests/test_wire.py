"""CPU tests (-m "not gpu") of the client wire format and admission rules (include/xlating_wire.h, SURVEY section
8(f) rank 4): byte layouts of src/api.h:13-38 (packed, big-endian), and the accept / reject decisions the reference's
own test/test_tcp_server.c expects from a server configured with band_sampling_rate 2 400 000 (test/resources/
tcp_server.config:3)."""
import struct

import sdr_server_amd as xl

BAND_RATE = 2400000
FILE, SOCKET = 0, 1


def test_message_bytes_are_the_packed_big_endian_structs():
    # request: header {version 0, type 0} + {u32 center, u32 rate, u32 band, u8 destination}, api.h:13-27
    msg = xl.wire_build_request(460700000, 48000, 460600000, FILE)
    assert msg == struct.pack(">BBIIIB", 0, 0, 460700000, 48000, 460600000, 0) and len(msg) == 15
    code, mtype = xl.wire_parse_header(msg[:2])
    assert (code, mtype) == (0, 0)
    code, req = xl.wire_parse_request(msg[2:])
    assert code == 0 and (req.center_freq, req.sampling_rate, req.band_freq, req.destination) == (460700000, 48000, 460600000, 0)
    # response: header {0, 2} + {u8 status, u32 details}, api.h:29-38, tcp_server.c:143-150
    r = xl.wire_build_response(0, 7)
    assert r == struct.pack(">BBBI", 0, 2, 0, 7) and len(r) == 7
    assert xl.wire_parse_response(r) == (0, 0, 7)
    assert xl.wire_parse_response(struct.pack(">BBBI", 0, 2, 1, 2)) == (0, 1, 2)
    # header-only messages (SHUTDOWN 1, PING 3)
    assert xl.wire_build_header(1) == b"\x00\x01" and xl.wire_build_header(3) == b"\x00\x03"
    # short reads and foreign protocol versions (tcp_server.c:91-94, 414-417)
    assert xl.wire_parse_header(b"\x00")[0] == -11  # EAGAIN
    assert xl.wire_parse_header(b"\x99\x00")[0] == -71  # EPROTO
    assert xl.wire_parse_request(msg[2:10])[0] == -11
    assert xl.wire_parse_response(b"\x00\x00\x00\x00\x00\x00\x00")[0] == -71  # not a RESPONSE


def admit(center, rate, band, dest, current_band=0):
    code, req = xl.wire_parse_request(xl.wire_build_request(center, rate, band, dest)[2:])
    assert code == 0
    return xl.wire_admit(req, BAND_RATE, current_band, 5)


def test_admission_matches_the_references_tcp_server_tests():
    # test_tcp_server.c:119-120 / 46-47: accepted
    code, adm, why = admit(460700000, 48000, 460600000, FILE)
    assert code == 0 and why == 0
    assert (adm.decimation, adm.center_offset, adm.lpf_cutoff, adm.lpf_transition) == (50, 100000, 24000, 9600)
    # test_tcp_server.c:77-113 test_invalid_request: every one answers INVALID_REQUEST (1)
    for center, rate, band, dest in ((460700000, 48000, 0, FILE),           # :80 missing band_freq
                                     (460700000, 0, 460600000, FILE),       # :84 missing sampling_rate
                                     (0, 48000, 460600000, FILE),           # :88 missing center_freq
                                     (460700000, 47000, 460600000, FILE),   # :100 not an integer factor of the band rate
                                     (462400000, 48000, 460600000, FILE),   # :104 above the band
                                     (458800000, 48000, 460600000, FILE),   # :108 below the band
                                     (460700000, 48000, 460600000, 0x99)):  # :112 unknown destination
        code, _, why = admit(center, rate, band, dest)
        assert code == -22 and why == 1, (center, rate, band, dest)
    # test_tcp_server.c:43-62: a second client on another band while the device runs -> OUT_OF_BAND_FREQ (2); accepted
    # again once nobody is running
    code, _, why = admit(460700000, 48000, 461600000, FILE, current_band=460600000)
    assert code == -22 and why == 2
    assert admit(460700000, 48000, 461600000, FILE, current_band=0)[0] == 0
    # band edges are inclusive (tcp_server.c:130, 136 use < and >)
    assert admit(460600000 + 1200000 - 24000, 48000, 460600000, SOCKET)[0] == 0
    assert admit(460600000 + 1200000 - 23999, 48000, 460600000, SOCKET)[0] == -22
    # test_tcp_server.c:161 test_rtlsdr: -12000 + 460100200 at 9600 -> D = 250, offset -12000 (dsp_worker.c:96-104)
    code, adm, _ = admit(460100200 - 12000, 9600, 460100200, SOCKET)
    assert code == 0 and (adm.decimation, adm.center_offset, adm.lpf_cutoff, adm.lpf_transition) == (250, -12000, 4800, 1920)
