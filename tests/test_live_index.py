"""The index phases' rules for LIVE streams (jsmpeg_amd/csrc/index_tables.h: JmStream::live_flags / live_limit,
JmPic::end_pos / mb_index), on the CPU through the test-only simulator build: which pictures of a stream that has only
PARTLY arrived a pass decodes, which ones it holds, and where the reference's cursor would rest (mpeg1.c:853-864,
980-984).  The GPU runs the same functions inside k_index (tests/test_gpu_live.py)."""
import ctypes
import glob
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from jsmpeg_amd import synth

HOLD, NONE = 1, 0xffffffff


@pytest.fixture(scope="module")
def sim():
    so = os.path.join(ROOT, "tests", "sim", "libjsmpeg_sim.so")
    src = os.path.join(ROOT, "tests", "sim", "sim_decode.cpp")
    csrc = os.path.join(ROOT, "jsmpeg_amd", "csrc")
    deps = [src] + glob.glob(os.path.join(csrc, "*.h"))
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I", csrc, "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.sim_index_live.restype = ctypes.c_int
    lib.sim_index_live.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                   ctypes.c_void_p] + [ctypes.c_void_p] * 5 + [ctypes.c_uint32, ctypes.c_void_p]
    return lib


def index(sim, es, w, h, flags, limit=0, known=None, want_record=False):
    cap = 4096
    pos, end = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    dec, mbi, fwd = np.zeros(cap, np.uint8), np.zeros(cap, np.uint32), np.zeros(cap, np.int32)
    hdr = np.zeros(4, np.int32)
    rec = np.zeros(sim.sim_stream_record_bytes(), np.uint8)
    es = np.ascontiguousarray(es, dtype=np.uint8)
    n = sim.sim_index_live(es.ctypes.data, len(es), w, h, flags, limit, known.ctypes.data if known is not None else None, rec.ctypes.data,
                           pos.ctypes.data, end.ctypes.data, dec.ctypes.data, mbi.ctypes.data, fwd.ctypes.data, cap, hdr.ctypes.data)
    out = dict(n=n, pos=pos[:n].tolist(), end=end[:n].tolist(), decoded=dec[:n].tolist(), mb_index=mbi[:n].tolist(), fwd=fwd[:n].tolist(),
               valid=int(hdr[0]), width=int(hdr[1]), height=int(hdr[2]), found=int(hdr[3]))
    return (out, rec) if want_record else out


def start_codes(es):
    b = np.asarray(es, dtype=np.uint8)
    at = np.flatnonzero((b[:-3] == 0) & (b[1:-2] == 0) & (b[2:-1] == 1))
    return [(int(i), int(b[i + 3])) for i in at]


def reference_rule(es, cut):
    """plain restatement: of the bytes es[:cut], the pictures a `while (next picture is complete) decode();` loop takes --
    a picture is complete when a start code that is not a slice's (nor extension / user data in front of the slices)
    follows its slices INSIDE the bytes"""
    codes = [(p, c) for p, c in start_codes(es[:cut])]
    pics, first_seq = [], None
    for k, (p, c) in enumerate(codes):
        if c == 0xB3 and first_seq is None:
            first_seq = k
        if c != 0x00:
            continue
        j = k + 1
        while j < len(codes) and codes[j][1] in (0xB5, 0xB2):
            j += 1
        while j < len(codes) and 0x01 <= codes[j][1] <= 0xAF:
            j += 1
        pics.append(dict(pos=p, k=k, end=codes[j][0] if j < len(codes) else None))
    return codes, first_seq, pics


@pytest.mark.parametrize("cfg", [dict(), dict(syntax_quirks=2), dict(syntax_quirks=1), dict(stuff_pictures=3)], ids=["plain", "skipped_pictures", "syntax_quirks", "stuffing"])
def test_hold_takes_exactly_the_complete_pictures(sim, cfg):
    W, H = 176, 144
    es, offs = synth.generate_config("cfg1_720p", n_frames=9, width=W, height=H, **cfg)
    full = index(sim, es, W, H, 0)
    assert full["valid"] == 1 and all(e != NONE for e in full["end"])
    rng = np.random.default_rng(5)
    cuts = sorted(set(rng.integers(1, len(es), 300).tolist() + [int(o) + d for o in offs for d in (-1, 0, 1, 2, 3, 4, 5, 12) if 0 < int(o) + d <= len(es)] + [len(es)]))
    for cut in cuts:
        codes, first_seq, pics = reference_rule(es, cut)
        got = index(sim, es[:cut], W, H, HOLD)
        assert got["n"] == len(pics)
        for i, pic in enumerate(pics):          # a picture the reference consumes without decoding (B / D / f_code 0, or before the header) ends with its header (mpeg1.c:955-967)
            if not full["decoded"][i]:
                pic["end"] = codes[pic["k"] + 1][0] if pic["k"] + 1 < len(codes) else None
        if first_seq is None or first_seq + 1 >= len(codes):
            # no header yet, or one that nothing ends: nothing is decoded; a begun header says where it begins
            assert not any(got["decoded"])
            assert (got["valid"], got["width"]) == ((-1, codes[first_seq][0]) if first_seq is not None else (0, 0))
            continue
        assert got["valid"] == 1 and got["found"] == 1
        for i, pic in enumerate(pics):
            if pic["end"] is None and i == len(pics) - 1:          # nothing ends the last picture yet: held
                assert got["end"][i] == NONE and got["decoded"][i] == 0
            else:
                # what the full stream's index says of that picture (a picture before the header is never decoded)
                assert got["end"][i] == full["end"][i] and got["decoded"][i] == full["decoded"][i], (cut, i)
        # the flushing form takes the last picture too, ending where the data ends
        fl = index(sim, es[:cut], W, H, 0)
        # (data that ends "00 00 01": the reference's scan takes that for a start code too, buffer.c:73-92, and rewinds onto it)
        data_end = cut - 3 if cut >= 3 and bytes(es[cut - 3:cut].tolist()) == b"\x00\x00\x01" else cut
        assert all(e != NONE for e in fl["end"]) and (not pics or not full["decoded"][len(pics) - 1] or fl["end"][-1] == (pics[-1]["end"] if pics[-1]["end"] is not None else data_end))


def test_limit_holds_the_rest_and_a_known_header_needs_no_header(sim):
    W, H = 176, 144
    es, offs = synth.generate_config("cfg1_720p", n_frames=9, width=W, height=H)
    full, rec = index(sim, es, W, H, 0, want_record=True)
    lim = index(sim, es, W, H, 0, limit=4)
    assert lim["decoded"] == [1, 1, 1, 1] + [0] * 5 and lim["end"][:4] == full["end"][:4] and all(e == NONE for e in lim["end"][4:])
    assert lim["mb_index"][:4] == [0, 1, 2, 3] and lim["fwd"][:4] == [-1, 0, 1, 2]
    # the next pass begins on the first held picture and has no header in its bytes: the record carries it
    at = lim["pos"][4]
    tail = index(sim, es[at:], W, H, HOLD, limit=4, known=rec)
    assert tail["valid"] == 1 and tail["found"] == 0
    assert tail["decoded"] == [1, 1, 1, 1, 0] and [e + at for e in tail["end"][:4]] == full["end"][4:8] and tail["end"][4] == NONE
    assert tail["fwd"][:4] == [-1, 0, 1, 2]                  # (the first P picture's reference is the ring's frame: live.hip seeds it)
    without = index(sim, es[at:], W, H, HOLD, limit=4)
    assert without["valid"] == 0 and not any(without["decoded"])
    # a header of another size makes the stream invalid: nothing is decoded
    other = index(sim, es, W + 16, H, 0)
    assert other["valid"] == 0 and other["found"] == 1 and not any(other["decoded"]) and (other["width"], other["height"]) == (W, H)
