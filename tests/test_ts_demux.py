"""Ingest side: MPEG-TS demux with the reference's semantics (src/ts.js:25-210).
The fixtures tests/golden/ts_*.json were agreed between the unmodified ts.js under Node and the CPU restatement
(tests/golden/make_golden_ts.py).  CPU tests pin the restatement; GPU tests compare the device demux (k_ts_* behind
jsmpeg_hip_batch_upload_ts) with both, then decode what it left in HBM."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import ts_craft
from conftest import ROOT
from jsmpeg_amd import cabi, hashing, synth
from oracle import checkers

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ts_*.json")))
IDS = [os.path.basename(p)[3:-5] for p in FIXTURES]


def load_case(path):
    fx = json.load(open(path))
    ts = ts_craft.CASES[fx["case"]]()
    assert hashlib.md5(ts.tobytes()).hexdigest() == fx["ts_md5"]
    return fx, ts


def as_fixture_writes(es, writes):
    return [dict(pts=p, length=n, md5=hashlib.md5(es[o:o + n].tobytes()).hexdigest()) for p, o, n in writes]


@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_oracle_matches_reference_fixture(path, libs):
    fx, ts = load_case(path)
    es, writes = checkers.oracle_ts_demux(libs["oracle"], ts, fx["stream_id"], fx.get("write_sizes"))
    assert as_fixture_writes(es, writes) == fx["writes"]


def test_oracle_feeds_the_decoder_like_ts_js(libs):
    """The demuxed bytes are the elementary stream: decoding them gives the stream's golden frames."""
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "frames_cfg0_240p_intra.json")))
    es, offs = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
    got, writes = checkers.oracle_ts_demux(libs["oracle"], synth.mux_ts(es, offs), 0xE0)
    assert len(writes) == fx["n_frames"]
    frames, _, _ = cabi.decode_stream(libs["oracle"], got)
    assert frames == fx["frame_md5"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_device_demux_matches_reference_fixture(path, hip_lib):
    """Every fixture -- aligned input, garbage in front of and between packets (resync), a partial last packet, and the
    buffers handed over in several write() calls (leftover bytes) -- through the ingest stage of the batch."""
    from jsmpeg_amd import batch as jb
    fx, ts = load_case(path)
    ws = fx.get("write_sizes")
    with jb.Batch(176, 144, 2, 64, 1 << 20) as b:
        # a second, shorter stream beside it (written in one piece)
        b.upload_ts([ts, ts[:188 * 7]], fx["stream_id"], None if ws is None else [ws, [188 * 7]])
        es = b.read_es(0)
        assert as_fixture_writes(es, b.ts_writes(0)) == fx["writes"]
        assert hashlib.md5(es.tobytes()).hexdigest() == fx["total_md5"]


def test_packet_framing_matches_the_restatement_on_random_damage(libs, hip_lib):
    """The host pre-pass of the ingest stage (csrc/ts_sync.h, jsmpeg_hip_ts_packet_runs) frames packets exactly where
    ts.js does -- by its CPU restatement, which the fixtures pin to ts.js itself: random junk (with and without sync
    bytes in it) spliced into a TS, truncated ends, random write() sizes: same packets, same leftover position."""
    import ctypes
    from jsmpeg_amd import batch as jb
    L = jb.lib()
    ora = ctypes.CDLL(libs["oracle"])
    ora.ts_oracle_packets.restype = ctypes.c_long
    rng = np.random.RandomState(5)
    base = ts_craft.case_video_audio_null()
    u64 = ctypes.c_uint64
    for trial in range(300):
        parts, at = [], 0
        for cut in sorted(rng.choice(np.arange(1, len(base) // 188), size=rng.randint(0, 4), replace=False)):
            parts.append(base[at:cut * 188])
            junk = rng.randint(0, 256, size=rng.randint(1, 1300)).astype(np.uint8)
            if trial % 3 == 0:
                junk[junk == 0x47] = 0x48
            parts.append(junk)
            at = cut * 188
        parts.append(base[at:len(base) - rng.randint(0, 400)])
        ts = np.ascontiguousarray(np.concatenate(parts))
        sizes, left = [], len(ts)
        if trial % 4:
            while left > 0:
                n = int(min(left, rng.randint(1, 4000)))
                sizes.append(n)
                left -= n
        ws = (u64 * max(1, len(sizes)))(*sizes)
        cap = len(ts) // 188 + 8
        want_at = (u64 * cap)()
        want_left = u64()
        n_want = ora.ts_oracle_packets(ctypes.c_void_p(ts.ctypes.data), ctypes.c_size_t(len(ts)), ws, ctypes.c_int(len(sizes)),
                                       want_at, ctypes.c_size_t(cap), ctypes.byref(want_left))
        ro, rp = (u64 * cap)(), (ctypes.c_uint32 * cap)()
        n_pk, left_at = u64(), u64()
        n_runs = L.jsmpeg_hip_ts_packet_runs(ctypes.c_void_p(ts.ctypes.data), u64(len(ts)), ws, ctypes.c_uint32(len(sizes)), ro, rp,
                                             ctypes.c_uint32(cap), ctypes.byref(n_pk), ctypes.byref(left_at))
        assert n_runs >= 0
        got = [ro[i] + 188 * k for i in range(n_runs) for k in range(rp[i])]
        assert n_pk.value == n_want == len(got), trial
        assert got == list(want_at[:n_want]), trial
        assert left_at.value == want_left.value, trial


@pytest.mark.gpu
def test_ts_in_planes_out(hip_lib, libs):
    """TS buffers -> device demux -> batch decode: every picture against the oracle fed by the CPU demux."""
    from jsmpeg_amd import batch as jb
    streams, want = [], []
    for s in range(5):
        es, offs = synth.generate_config("cfg1_720p", n_frames=13, stream=s, width=352, height=288)
        ts = synth.mux_ts(es, offs)
        streams.append(ts)
        demuxed, writes = checkers.oracle_ts_demux(libs["oracle"], ts, 0xE0)
        assert len(writes) == 13
        frames, _, _ = cabi.decode_stream(libs["oracle"], demuxed, keep="planes")
        want.append([hashing.frame_hash(*f) for f in frames])
    with jb.Batch(352, 288, 5, 5 * 13 + 4, 1 << 22) as b:
        b.upload_ts(streams)
        for s in range(5):
            w = b.ts_writes(s)
            assert len(w) == 13 and abs(w[1][0] - w[0][0] - 1 / 30) < 1e-4
        n = b.decode()
        assert n == 5 * 13
        dev = b.frame_hashes()
        per_stream = {}
        for p, info in enumerate(b.pictures()):
            per_stream.setdefault(info.stream, []).append(int(dev[p]))
        for s in range(5):
            assert per_stream[s] == want[s], "stream %d" % s


def host_demux(L, ts, stream_id, write_sizes=None):
    import ctypes
    ts = np.ascontiguousarray(ts, dtype=np.uint8)
    es = np.zeros(len(ts) + 16, dtype=np.uint8)
    cap = len(ts) // 94 + 16
    pts, off, ln = np.zeros(cap, np.float64), np.zeros(cap, np.uint64), np.zeros(cap, np.uint32)
    n_es = ctypes.c_uint64()
    ws = (ctypes.c_uint64 * max(1, len(write_sizes or [])))(*[int(x) for x in (write_sizes or [])])
    fn = L.jsmpeg_hip_ts_demux_host
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                   ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    n = fn(ts.ctypes.data, len(ts), ws, len(write_sizes or []), stream_id, es.ctypes.data, len(es), ctypes.byref(n_es), pts.ctypes.data, off.ctypes.data, ln.ctypes.data, cap)
    assert 0 <= n <= cap
    return es[:n_es.value].copy(), [(float(pts[i]), int(off[i]), int(ln[i])) for i in range(n)]


@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_live_streams_host_demuxer_matches_reference_fixture(path, hip_lib):
    """the demuxer a LIVE stream keeps in front of its write() (jsmpeg_hip_live_write_ts: ts.js's state carried between calls --
    leftover bytes, resync, the PES being collected), run by itself on the host (jsmpeg_hip_ts_demux_host): the same
    destination.write calls as the unmodified ts.js under Node, for every fixture and its write() sizes"""
    from jsmpeg_amd import batch as jb
    fx, ts = load_case(path)
    es, writes = host_demux(jb.lib(), ts, fx["stream_id"], fx.get("write_sizes"))
    assert as_fixture_writes(es, writes) == fx["writes"]
    assert hashlib.md5(es.tobytes()).hexdigest() == fx["total_md5"]


def test_live_streams_host_demuxer_on_random_damage_and_pieces(libs, hip_lib):
    """... and against the restatement (pinned to ts.js by the fixtures) on random junk spliced into a TS, truncated ends and
    random write() sizes: same bytes, same write() boundaries, same pts"""
    from jsmpeg_amd import batch as jb
    rng = np.random.RandomState(9)
    base = ts_craft.case_video_audio_null()
    for trial in range(200):
        parts, at = [], 0
        for cut in sorted(rng.choice(np.arange(1, len(base) // 188), size=rng.randint(0, 4), replace=False)):
            parts.append(base[at:cut * 188])
            junk = rng.randint(0, 256, size=rng.randint(1, 1300)).astype(np.uint8)
            if trial % 3 == 0:
                junk[junk == 0x47] = 0x48
            parts.append(junk)
            at = cut * 188
        parts.append(base[at:len(base) - rng.randint(0, 400)])
        ts = np.ascontiguousarray(np.concatenate(parts))
        sizes, left = [], len(ts)
        if trial % 4:
            while left > 0:
                n = int(min(left, rng.randint(1, 4000)))
                sizes.append(n)
                left -= n
        sid = 0xE0 if trial % 5 else 0xC0
        want_es, want_w = checkers.oracle_ts_demux(libs["oracle"], ts, sid, sizes or None)
        got_es, got_w = host_demux(jb.lib(), ts, sid, sizes or None)
        assert got_w == want_w, trial
        assert np.array_equal(got_es, want_es[:len(got_es)]) and len(got_es) == sum(w[2] for w in want_w), trial
