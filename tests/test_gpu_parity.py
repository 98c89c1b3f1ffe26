"""Parity tests proper: the HIP path (through the C ABI in libjsmpeg_hip.so)
against the committed golden fixtures and against the oracle on the same
seeded inputs.  Bit-exact: the path is integer/byte work.  Needs an MI355X."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from jsmpeg_amd import batch as jb
from jsmpeg_amd import cabi, hashing, synth

pytestmark = pytest.mark.gpu

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "frames_*.json")))
IDS = [os.path.basename(p)[7:-5] for p in FIXTURES]


def load_case(path):
    fx = json.load(open(path))
    es, offs = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
    assert hashlib.md5(es.tobytes()).hexdigest() == fx["es_md5"]
    return fx, es, offs


def md5_planes(planes):
    h = hashlib.md5()
    for p in planes:
        h.update(p.tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_batch_matches_golden(path, hip_lib):
    fx, es, _ = load_case(path)
    with jb.Batch(fx["info"]["width"], fx["info"]["height"], 1, len(fx.get("abi_frame_md5", fx["frame_md5"])) + 2, len(es) + 1024) as b:
        b.upload([es])
        n = b.decode()
        pics = b.pictures()
        # pictures the reference consumes without decoding (B / D / f_code 0) are listed, flagged, and have no frame
        assert n == len(fx.get("abi_frame_md5", fx["frame_md5"]))
        decoded = [p for p in range(n) if pics[p].decoded]
        assert len(decoded) == fx["n_frames"]
        got = [md5_planes(b.read_frame(p)) for p in decoded]
        assert got == fx["frame_md5"]
        # device-side hash agrees with the host mirror on the copied-back planes
        dev = b.frame_hashes()
        for p in (decoded[0], decoded[len(decoded) // 2], decoded[-1]):
            assert int(dev[p]) == hashing.frame_hash(*b.read_frame(p))
        # every picture in ONE strided copy into pinned memory (jsmpeg_hip_batch_read_frames) == picture by picture
        allf = b.read_frames(0, n)
        lu, ch = b.luma_bytes, b.chroma_bytes
        assert [md5_planes((allf[p, :lu], allf[p, lu:lu + ch], allf[p, lu + ch:])) for p in decoded] == fx["frame_md5"]
        tail = b.read_frames(decoded[-1], 1).copy()
        assert md5_planes((tail[0, :lu], tail[0, lu:lu + ch], tail[0, lu + ch:])) == fx["frame_md5"][-1]
        with pytest.raises(RuntimeError):
            b.read_frames(n, 1)


@pytest.mark.parametrize("split", ["0", "1"], ids=["k_parse", "k_parse_split"])
def test_both_forms_of_the_parse_match_golden(split, hip_lib, monkeypatch):
    """jm_launch_parse picks the kernel per pass by the content's bytes per macroblock (kernels.hip): the ring service in one
    piece (k_parse) or in two halves a turn apart (k_parse_split).  Here every fixture goes through each, forced."""
    monkeypatch.setenv("JSMPEG_HIP_PARSE_SPLIT", split)
    for path in FIXTURES:
        fx, es, _ = load_case(path)
        with jb.Batch(fx["info"]["width"], fx["info"]["height"], 1, len(fx.get("abi_frame_md5", fx["frame_md5"])) + 2, len(es) + 1024) as b:
            b.upload([es])
            n = b.decode()
            pics = b.pictures()
            got = [md5_planes(b.read_frame(p)) for p in range(n) if pics[p].decoded]
            assert got == fx["frame_md5"], os.path.basename(path)


@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_decoder_abi_matches_golden(path, hip_lib):
    """The reference's 15-function ABI, one write, pull every picture."""
    fx, es, offs = load_case(path)
    frames, idx, info = cabi.decode_stream(hip_lib, es)
    assert frames == fx.get("abi_frame_md5", fx["frame_md5"])
    assert idx == fx["bit_index_after_decode"]
    assert info["coded_size"] == fx["info"]["coded_size"]
    assert info["width"] == fx["info"]["width"] and info["height"] == fx["info"]["height"]
    assert abs(info["frame_rate"] - fx["info"]["frame_rate"]) < 1e-6


def test_decoder_abi_streaming_evict(hip_lib):
    """EVICT store, a picture written / a picture pulled (how ts.js + Player drive it)."""
    fx, es, offs = load_case(os.path.join(ROOT, "tests", "golden", "frames_cfg0_240p_intra.json"))
    got = []
    with cabi.Mpeg1Decoder(hip_lib, 24 * 1024, cabi.MODE_EVICT) as dec:
        n = len(offs) - 1
        for k in range(n):
            end = len(es) if k == n - 1 else int(offs[k + 1])
            dec.write(es[int(offs[k]):end])
            while dec.decode():
                got.append(md5_planes(dec.planes()))
    assert got == fx["frame_md5"]


def test_decoder_abi_per_picture_writes_expand(hip_lib):
    fx, es, offs = load_case(os.path.join(ROOT, "tests", "golden", "frames_long_gop_p_chain.json"))
    frames, _, _ = cabi.decode_stream(hip_lib, es, offs, buffer_size=4096)  # forces the store to grow
    assert frames == fx["frame_md5"]


def test_batch_many_streams_vs_oracle(hip_lib, libs):
    """8 streams with distinct seeds in one batch; every frame against the oracle."""
    streams, want = [], []
    for s in range(8):
        es, _ = synth.generate_config("cfg1_720p", n_frames=13, stream=s, width=352, height=288)
        streams.append(es)
        frames, _, _ = cabi.decode_stream(libs["oracle"], es, keep="planes")
        want.append([hashing.frame_hash(*f) for f in frames])
    with jb.Batch(352, 288, 8, 8 * 13 + 4, sum(len(s) for s in streams) + 4096) as b:
        b.upload(streams)
        n = b.decode()
        assert n == 8 * 13
        dev = b.frame_hashes()
        counters = b.counters()
        assert counters["levels"] == 12 and counters["decoded"] == n
        per_stream = {}
        for p, info in enumerate(b.pictures()):
            per_stream.setdefault(info.stream, []).append(int(dev[p]))
        for s in range(8):
            assert per_stream[s] == want[s], "stream %d" % s
        # decode again into the same batch object: epochs, not memsets, invalidate old records
        b.decode()
        assert np.array_equal(b.frame_hashes(), dev)


def test_randomised_sweep(hip_lib, libs):
    """tools/fuzz_parity.py, a short run: random sizes and generator parameters (also the unusual-syntax option), batch +
    one-picture ABI + device RGBA against the oracle."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "16", "3"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "16 cases, 0 mismatches" in out.stdout


def test_randomised_call_patterns_of_the_one_picture_interface(hip_lib, libs):
    """tools/fuzz_abi_chunks.py, a short run: random chunking, EVICT stores that evict, EXPAND stores that grow, partial pulls
    that leave pictures decoded ahead queued across writes and evictions, seeks -- the identical call sequence on the product
    and on the oracle: every decode()'s return value, cursor and planes."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_abi_chunks.py"), "120", "5"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "120 cases, 0 mismatches" in out.stdout
    assert int(out.stdout.rsplit(";", 1)[1].split()[0]) > 200      # ... and the decode-ahead was part of it


def test_cfg4_2160p_batch_at_scale(hip_lib, libs):
    """SURVEY.md 8d cfg4 (3840x2160, high bitrate): 6 streams x 13 pictures in one batch, every frame hash against the
    oracle (decoded on host threads)."""
    import threading
    streams, want = [], [None] * 6
    for s in range(6):
        es, _ = synth.generate_config("cfg4_2160p", n_frames=13, stream=s)
        streams.append(es)

    def ref(s):
        frames, _, _ = cabi.decode_stream(libs["oracle"], streams[s], keep="planes")
        want[s] = [hashing.frame_hash(*f) for f in frames]

    ts = [threading.Thread(target=ref, args=(s,)) for s in range(6)]
    [t.start() for t in ts]
    with jb.Batch(3840, 2160, 6, 6 * 13 + 4, sum(len(s) for s in streams) + 8192) as b:
        b.upload(streams)
        assert b.decode() == 6 * 13
        dev = b.frame_hashes()
        per = {}
        for p, info in enumerate(b.pictures()):
            per.setdefault(info.stream, []).append(int(dev[p]))
    [t.join() for t in ts]
    for s in range(6):
        assert per[s] == want[s], "stream %d" % s


def test_damaged_input_neither_faults_nor_hangs(hip_lib):
    """tools/fuzz_corrupt.py, a short run: byte damage, truncation, spliced garbage, sprinkled start codes, damaged TS
    packets.  Like the reference, no error is reported for bitstream content; unlike it, nothing is read or written out of bounds."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_corrupt.py"), "40", "2"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "40 damaged streams decoded without fault or hang" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("env", [dict(JSMPEG_HIP_PARSE_RESIDENT="1", JSMPEG_HIP_PARSE_LANES="64"),
                                 dict(JSMPEG_HIP_PARSE_RESIDENT="2", JSMPEG_HIP_PARSE_LANES="4"),
                                 dict(JSMPEG_HIP_PARSE_HEAD="5,1,23,2"), dict(JSMPEG_HIP_PARSE_HEAD="64,4,1000,8"),
                                 dict(JSMPEG_HIP_PARSE_HEAD="9,3,9,3", JSMPEG_HIP_PARSE_RESIDENT="1")])
def test_parse_wavefronts_draw_batches_by_ticket(env):
    """Large passes launch as many parse workgroups as the GPU holds and their wavefronts draw further batches of slices
    from a ticket counter; small inputs never get there -- here they do (the environment limits the workgroups: the
    library reads it once per process, hence the subprocess), golden fixtures through the batch interface.
    JSMPEG_HIP_PARSE_HEAD: the batches of the longest slices take fewer slices per wavefront than the rest (what
    jm_launch_parse does by itself for passes with a few long slices) -- here forced on every fixture, in segmentations
    that do not divide the slice count."""
    import subprocess
    import sys
    code = r'''
import glob, hashlib, json, os, sys
sys.path.insert(0, %r)
import numpy as np
from jsmpeg_amd import batch as jb, synth
bad = []
for path in sorted(glob.glob(os.path.join(%r, "tests", "golden", "frames_*.json"))):
    fx = json.load(open(path))
    if fx["n_frames"] * fx["info"]["coded_size"] > 40e6 or "abi_frame_md5" in fx:
        continue
    es, _ = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
    with jb.Batch(fx["info"]["width"], fx["info"]["height"], 2, 2 * fx["n_frames"] + 4, 2 * len(es) + 8192) as b:
        b.upload([es, es])
        assert b.decode() == 2 * fx["n_frames"]
        for p in range(2 * fx["n_frames"]):
            h = hashlib.md5()
            for plane in b.read_frame(p):
                h.update(plane.tobytes())
            if h.hexdigest() != fx["frame_md5"][p %% fx["n_frames"]]:
                bad.append((os.path.basename(path), p))
print("BAD", bad)
assert not bad
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


# ---- decode-ahead in the one-picture ABI (include/jsmpeg_hip.h): the batch engine behind decode(), invisible to the caller ----

def test_decode_ahead_serves_buffered_streams_and_is_invisible(hip_lib, libs):
    """a buffered stream (everything written, decode() until false): the pictures come from passes of the batch engine --
    and planes, cursor after every call and picture count are what the oracle gives one picture at a time.  Then seeks
    (set_index back onto earlier pictures, in the middle of a pass) against the oracle doing the same."""
    fx, es, offs = load_case(os.path.join(ROOT, "tests", "golden", "frames_cfg1_720p.json"))
    with cabi.Mpeg1Decoder(hip_lib, len(es) + 1024, cabi.MODE_EXPAND) as d, cabi.Mpeg1Decoder(libs["oracle"], len(es) + 1024, cabi.MODE_EXPAND) as o:
        d.write(es)
        o.write(es)
        n = 0
        while True:
            a, b = d.decode(), o.decode()
            assert a == b
            if not a:
                break
            assert d.index == o.index and md5_planes(d.planes()) == md5_planes(o.planes()), n
            n += 1
        assert n == fx["n_frames"]
        passes, served = d.ahead_stats()
        assert passes >= 1 and served >= n - 3            # the first picture and a last one with nothing behind it come the plain way
        # seeks: back to picture 7 (a pass is dropped half way), on for five pictures, back to picture 3, to the end
        for target, count in ((7, 5), (3, 1000), (12, 2), (0, 3)):
            d.index = int(offs[target]) * 8
            o.index = int(offs[target]) * 8
            for k in range(count):
                a, b = d.decode(), o.decode()
                assert a == b
                if not a:
                    break
                assert d.index == o.index and md5_planes(d.planes()) == md5_planes(o.planes()), (target, k)
        assert d.ahead_stats()[0] > passes


def test_decode_ahead_stays_out_of_streaming_and_can_be_switched_off(hip_lib):
    fx, es, offs = load_case(os.path.join(ROOT, "tests", "golden", "frames_long_gop_p_chain.json"))
    # streaming: a picture written, a picture pulled -- nothing is ever buffered ahead
    with cabi.Mpeg1Decoder(hip_lib, 64 * 1024, cabi.MODE_EVICT) as dec:
        got, n = [], len(offs) - 1
        for k in range(n):
            end = len(es) if k == n - 1 else int(offs[k + 1])
            dec.write(es[int(offs[k]):end])
            while dec.decode():
                got.append(md5_planes(dec.planes()))
        assert got == fx["frame_md5"] and dec.ahead_stats() == (0, 0)
    # EVICT with several pictures per write: passes of the batch engine, the store evicting underneath them
    with cabi.Mpeg1Decoder(hip_lib, 48 * 1024, cabi.MODE_EVICT) as dec:
        got, n = [], len(offs) - 1
        for k in range(0, n, 5):
            hi = min(n, k + 5)
            end = len(es) if hi == n else int(offs[hi])
            dec.write(es[int(offs[k]):end])
            while dec.decode():
                got.append(md5_planes(dec.planes()))
        assert got == fx["frame_md5"] and dec.ahead_stats()[1] > 0
    os.environ["JSMPEG_HIP_DECODE_AHEAD"] = "0"
    try:
        frames, idx, _ = cabi.decode_stream(hip_lib, es)
    finally:
        del os.environ["JSMPEG_HIP_DECODE_AHEAD"]
    assert frames == fx["frame_md5"] and idx == fx["bit_index_after_decode"]


def test_index_chain_as_scans_over_long_streams(hip_lib, libs):
    """k_index works a stream's forward references and dependency levels out as workgroup scans, 256 pictures at a time with
    carries (kernels.hip); index_tables.h jm_index_chain is the definition.  Streams of 300-700 small pictures -- GOPs of 1 to
    400, pictures the reference consumes without decoding (B / D, f_code 0) across the chunk boundaries, a stream whose first
    decoded picture is a P picture -- : the table's forward / level fields == the chain restated here, pictures == the oracle"""
    W, H = 64, 48
    specs = [dict(n_frames=300, gop=1), dict(n_frames=700, gop=400), dict(n_frames=520, gop=7, syntax_quirks=2),
             dict(n_frames=513, gop=256, syntax_quirks=3), dict(n_frames=257, gop=12)]
    streams = [synth.generate_config("cfg1_720p", stream=300 + i, width=W, height=H, **kw)[0] for i, kw in enumerate(specs)]
    # a stream that BEGINS with a P picture: the sequence header, then the stream from its second picture on
    es, offs = synth.generate_config("cfg1_720p", stream=309, width=W, height=H, n_frames=300, gop=300)
    offs = [int(o) for o in offs]
    streams.append(np.concatenate([es[:offs[0]], es[offs[1]:]]) if offs[0] > 0 else es)
    n_max = sum(s.tobytes().count(b"\x00\x00\x01\x00") for s in streams) + 16
    with jb.Batch(W, H, len(streams), n_max, sum(len(s) for s in streams) + 65536) as b:
        b.upload(streams)
        n = b.decode()
        pics = b.pictures()
        last = {}
        for p in range(n):
            info = pics[p]
            if not info.decoded:
                continue
            prev = last.get(info.stream)
            if info.type == 2 and prev is not None:
                want = (prev[0], prev[1] + 1)
            else:
                want = (-1, 0)
            assert (info.forward, info.level) == want, (p, info.stream, info.type, info.forward, info.level, want)
            last[info.stream] = (p, want[1])
        assert max(pics[p].level for p in range(n) if pics[p].decoded) >= 399
        dev = b.frame_hashes()
        for s, es_s in enumerate(streams):
            frames, _, _ = cabi.decode_stream(libs["oracle"], es_s, keep="planes")
            want_h = [hashing.frame_hash(*f) for f in frames]
            got_h = [int(dev[p]) for p in range(n) if pics[p].stream == s]
            # (pictures the reference consumes without decoding repeat the planes before them in the oracle's list)
            dec = [int(dev[p]) for p in range(n) if pics[p].stream == s and pics[p].decoded]
            dedup = [x for k, x in enumerate(want_h) if k == 0 or x != want_h[k - 1]]
            dd = [x for k, x in enumerate(dec) if k == 0 or x != dec[k - 1]]
            assert dd == dedup, (s, len(dd), len(dedup))
            assert len(got_h) >= len(dec)
