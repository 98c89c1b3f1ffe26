"""LIVE streams (include/jsmpeg_hip.h part 5): streams that persist across calls, every pending picture of every stream in
one pass of the batch engine per tick -- against the golden fixtures and against the oracle fed THE SAME write() calls
(reference src/ts.js:205-210 -> decoder.js:36-47 -> buffer.js:64-104 -> mpeg1.c:853-864, 986-994).  Bit-exact.  Needs an MI355X."""
import glob
import hashlib
import json
import os
import random

import numpy as np
import pytest

from conftest import ROOT
from jsmpeg_amd import cabi, hashing, live as jl, synth

pytestmark = pytest.mark.gpu

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "frames_*.json")))
IDS = [os.path.basename(p)[7:-5] for p in FIXTURES]


def load_case(path):
    fx = json.load(open(path))
    es, offs = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
    assert hashlib.md5(es.tobytes()).hexdigest() == fx["es_md5"]
    return fx, es, [int(o) for o in offs]


def md5_planes(planes):
    h = hashlib.md5()
    for p in planes:
        h.update(p.tobytes())
    return h.hexdigest()


def picture_writes(es, offs):
    """the elementary stream as the reference's demuxer hands it to a decoder: one write per picture (ts.js:205-210)"""
    n = len(offs) - 1
    return [es[offs[k]:(len(es) if k == n - 1 else offs[k + 1])] for k in range(n)]


def drain(lv, got, per_stream=None):
    """the last tick's pictures: md5 of the planes read back (and the device hash agrees with the host's)"""
    pics = lv.pictures()
    dev = lv.frame_hashes()
    for i, p in enumerate(pics):
        planes = lv.read_frame(i)
        assert int(dev[i]) == hashing.frame_hash(*planes)
        (got if per_stream is None else per_stream.setdefault(p.stream, [])).append(md5_planes(planes))
    return pics


@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_live_picture_by_picture_matches_golden(path, hip_lib):
    """a picture written, a tick: how ts.js + Player.updateForStreaming drive a decoder -- every P picture predicts from a
    frame an EARLIER tick left in the stream's ring, unwritten macroblocks show the ring's frame before that"""
    fx, es, offs = load_case(path)
    got = []
    with jl.Live(fx["info"]["width"], fx["info"]["height"], 2, pictures_per_tick=2, store_bytes=max(1 << 16, 2 * len(es))) as lv:
        s = lv.open()
        for k, w in enumerate(picture_writes(es, offs)):
            lv.write(s, w, pts=k / 30.0)
            lv.tick(flush=True)
            drain(lv, got)
        info = lv.stream_info(s)
        assert info.has_sequence_header and (info.width, info.height) == (fx["info"]["width"], fx["info"]["height"])
        assert abs(info.frame_rate - fx["info"]["frame_rate"]) < 1e-6
        assert info.pending_bytes == 0 and info.pictures == fx["n_frames"] and info.evictions == 0
    assert got == fx["frame_md5"]


@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_live_ragged_chunks_match_golden(path, hip_lib):
    """bytes in arbitrary pieces, ticks that decode only COMPLETE pictures (flags 0), a FLUSH tick at the end: the
    pictures of the whole stream decoded in one piece, whatever the pieces were -- start codes, picture headers and
    slices cut anywhere, pieces of a few bytes and pieces of several pictures"""
    fx, es, offs = load_case(path)
    if fx["info"]["width"] * fx["info"]["height"] > 1280 * 720:
        pytest.skip("the small fixtures cover the cuts; the large ones go picture by picture above")
    rng = random.Random(len(es))
    mean = max(64, len(es) // (3 * fx["n_frames"]))
    got, at, ticks = [], 0, 0
    with jl.Live(fx["info"]["width"], fx["info"]["height"], 1, pictures_per_tick=3, store_bytes=2 * len(es) + 4096) as lv:
        s = lv.open()
        while at < len(es):
            n = min(len(es) - at, rng.choice([1, 2, 3, 5, 17, mean // 2, mean, mean, 2 * mean, 7 * mean]))
            lv.write(s, es[at:at + n])
            at += n
            if rng.random() < 0.7:
                lv.tick(flush=False)
                drain(lv, got)
                ticks += 1
        for _ in range(fx["n_frames"]):          # (at most three picture start codes per tick)
            lv.tick(flush=False)
            drain(lv, got)
        assert got == fx["frame_md5"][:len(got)] and len(got) >= fx["n_frames"] - 1, (len(got), fx["n_frames"])
        lv.tick(flush=True)                      # the last picture: nothing ends it but the end of the data
        drain(lv, got)
        assert lv.stream_info(s).pending_bytes == 0
    assert got == fx["frame_md5"] and ticks > 0


def test_live_three_pictures_at_a_time(hip_lib):
    fx, es, offs = load_case(os.path.join(ROOT, "tests", "golden", "frames_long_gop_p_chain.json"))
    writes = picture_writes(es, offs)
    got, counts = [], []
    with jl.Live(fx["info"]["width"], fx["info"]["height"], 1, pictures_per_tick=3, store_bytes=1 << 20) as lv:
        s = lv.open()
        for k in range(0, len(writes), 3):
            for j, w in enumerate(writes[k:k + 3]):
                lv.write(s, w, pts=(k + j) / 25.0)
            counts.append(lv.tick(flush=True))
            pics = drain(lv, got)
            assert [round(p.pts * 25) for p in pics] == list(range(k, min(k + 3, len(writes))))   # a picture carries its write's pts
            assert [p.type for p in pics] == [1 if (k + j) % 40 == 0 else 2 for j in range(len(pics))]
    assert got == fx["frame_md5"] and counts == [3] * (len(writes) // 3) + ([len(writes) % 3] if len(writes) % 3 else [])


def test_live_picture_limit_per_tick(hip_lib):
    """more pictures buffered than a tick takes: the rest wait, in order, whether the tick flushes or not"""
    fx, es, offs = load_case(os.path.join(ROOT, "tests", "golden", "frames_custom_quant_escapes.json"))
    got, counts = [], []
    with jl.Live(fx["info"]["width"], fx["info"]["height"], 1, pictures_per_tick=2, store_bytes=2 * len(es)) as lv:
        s = lv.open()
        lv.write(s, es)
        for _ in range(fx["n_frames"]):
            n = lv.tick(flush=True)
            if n == 0:
                break
            counts.append(n)
            drain(lv, got)
    assert got == fx["frame_md5"] and counts == [2] * (fx["n_frames"] // 2) + [1] * (fx["n_frames"] % 2)


def oracle_fed(libs, writes_per_tick, store_bytes):
    """the oracle's decoder (EVICT store, the reference's streaming mode) fed the same writes, `while (decode());` per tick"""
    out, last = [], None
    with cabi.Mpeg1Decoder(libs["oracle"], store_bytes, cabi.MODE_EVICT) as dec:
        for writes in writes_per_tick:
            for w in writes:
                dec.write(w)
            tick = []
            while dec.decode():
                # decode() is also true for a picture the reference consumes WITHOUT decoding (B / D pictures, f_code 0:
                # mpeg1.c:955-967): the planes are the same as before -- not a picture of the live interface, which lists what
                # was decoded.  (Two decoded pictures in a row with identical planes do not occur in this content.)
                h = md5_planes(dec.planes())
                if h != last:
                    tick.append(h)
                last = h
            out.append(tick)
    return out


def test_live_eight_streams_join_and_leave(hip_lib, libs):
    """8 streams with distinct content in one live batch, fed picture by picture; stream 5 joins at tick 4 (mid-run: the
    others are in the middle of their GOPs), stream 2 leaves at tick 9 and its id is taken by a NEW stream at tick 11
    (a fresh decoder: no header, zeroed planes); some ticks bring a stream nothing, some two pictures.  Every picture of
    every tick == the oracle fed the same writes."""
    W, H, N = 352, 288, 16
    def content(seed, **kw):
        es, offs = synth.generate_config("cfg1_720p", n_frames=N, stream=seed, width=W, height=H, **kw)
        return picture_writes(es, [int(o) for o in offs])
    plans = {s: content(100 + s) for s in range(8)}
    plans[3] = content(103, mv_jitter=1, f_code_max=1, coded_permille=60, ac_max=1, gop=2)          # unwritten last macroblocks (coherent pan)
    plans[6] = content(106, syntax_quirks=2)                                                         # B / D pictures, f_code 0 between the decoded ones
    newcomer = content(999, gop=4)
    join = {s: 0 for s in range(8)}
    join[5] = 4
    rng = random.Random(7)
    # per tick and stream: how many of its pictures arrive (0, 1 or 2)
    ticks = []
    given = {s: 0 for s in list(range(8)) + ["new"]}
    for t in range(3 * N):
        row = {}
        for s in range(8):
            if t < join[s] or (s == 2 and t >= 9):
                continue
            k = rng.choice([0, 1, 1, 1, 2])
            row[s] = plans[s][given[s]:given[s] + k]
            given[s] += len(row[s])
        if t >= 11:
            k = rng.choice([1, 1, 2])
            row["new"] = newcomer[given["new"]:given["new"] + k]
            given["new"] += len(row["new"])
        ticks.append(row)
    store = 1 << 18
    want = {s: oracle_fed(libs, [row.get(s, []) for row in ticks], store) for s in list(range(8)) + ["new"]}
    with jl.Live(W, H, 8, pictures_per_tick=6, store_bytes=store) as lv:      # (the limit counts picture START CODES: stream 6's B / D pictures too)
        ids = {}
        for t, row in enumerate(ticks):
            for s in range(8):
                if join[s] == t:
                    ids[s] = lv.open()
            if t == 9:
                lv.close_stream(ids.pop(2))
            if t == 11:
                ids["new"] = lv.open()
                assert ids["new"] == 2                      # the id that was given back
            for s, writes in row.items():
                for w in writes:
                    lv.write(ids[s], w, pts=float(t))
            lv.tick(flush=True)
            per = {}
            pics = drain(lv, None, per)
            assert [p.stream for p in pics] == sorted(p.stream for p in pics)
            assert all(p.pts == float(t) for p in pics)
            for s, i in ids.items():
                assert per.get(i, []) == want[s][t], (t, s)
        assert sum(len(x) for x in want[3]) > 0 and sum(len(x) for x in want["new"]) > 0


def test_live_store_evicts_like_the_reference(hip_lib, libs):
    """writes that outrun the ticks: a write that does not fit beside the UNDECODED bytes throws those away (buffer.js:37-56,
    the 'emergency evac') and the decoder goes on with what it has -- a P picture whose reference was lost predicts from
    the frame before, exactly as the reference does with the same writes"""
    W, H = 352, 288
    es, offs = synth.generate_config("cfg1_720p", n_frames=14, stream=11, width=W, height=H)
    writes = picture_writes(es, [int(o) for o in offs])
    store = int(max(len(w) for w in writes) * 2.5)
    # a tick after picture 0, then pictures 1 .. 5 without a tick (the store holds two and a half), a tick, the rest one by one
    ticks = [[writes[0]], writes[1:6]] + [[w] for w in writes[6:]]
    want = oracle_fed(libs, ticks, store)
    got = []
    with jl.Live(W, H, 1, pictures_per_tick=4, store_bytes=store) as lv:
        s = lv.open()
        for ws in ticks:
            for w in ws:
                lv.write(s, w)
            lv.tick(flush=True)
            tick = []
            drain(lv, tick)
            got.append(tick)
        assert lv.stream_info(s).evictions >= 1
        with pytest.raises(RuntimeError):
            lv.write(s, np.zeros(store + 1, dtype=np.uint8))
    assert got == want and sum(len(t) for t in got) < 14


def test_live_writes_beside_a_tick_in_flight(hip_lib, libs):
    """the tick in two halves (jsmpeg_hip_live_tick_begin / _end): the writes of tick t + 1 are made while tick t is on the
    device.  To the streams they are writes made right behind the tick -- so every tick's pictures == the oracle fed the
    writes tick by tick, with stores small enough that writes evict (room is decided when the tick has ended, not when the
    bytes were copied), a stream that joins mid-run (open() ends the tick first) and one whose first bytes are not video"""
    W, H, N = 352, 288, 18
    rng = random.Random(23)
    plans = {}
    for s in range(4):
        es, offs = synth.generate_config("cfg1_720p", n_frames=N, stream=40 + s, width=W, height=H, **({"syntax_quirks": 2} if s == 2 else {}))
        plans[s] = picture_writes(es, [int(o) for o in offs])
    plans[1] = [np.frombuffer(bytes(rng.randrange(256) for _ in range(700)), dtype=np.uint8)] + plans[1]   # noise before the first header
    store = int(max(len(w) for ws in plans.values() for w in ws) * 1.2) + 64
    join = {0: 0, 1: 0, 2: 0, 3: 5}
    given = {s: 0 for s in plans}
    ticks = []
    while any(given[s] < len(plans[s]) for s in plans):
        row = {}
        for s in plans:
            if len(ticks) < join[s]:
                continue
            k = rng.choice([0, 1, 1, 2, 3, 5])                    # several pictures into a store of 1.2 of the largest: evacuations
            row[s] = plans[s][given[s]:given[s] + k]
            given[s] += len(row[s])
        ticks.append(row)
    want = {s: oracle_fed(libs, [row.get(s, []) for row in ticks], store) for s in plans}
    got = {s: [] for s in plans}
    with jl.Live(W, H, 4, pictures_per_tick=8, store_bytes=store) as lv:
        ids = {}
        def feed(t):
            for s in plans:
                if join[s] == t:
                    ids[s] = lv.open()                            # (between the halves: ends the tick, its pictures stay readable)
            for s, ws in ticks[t].items():
                for w in ws:
                    lv.write(ids[s], w, pts=float(t))
        def collect(t):
            per = {}
            pics = drain(lv, None, per)
            assert all(p.pts == float(t) for p in pics)
            for s, i in ids.items():
                if join[s] <= t:
                    got[s].append(per.get(i, []))
        feed(0)
        for t in range(len(ticks)):
            lv.tick_begin(flush=True)
            if t + 1 < len(ticks):
                feed(t + 1)                                       # while tick t is on the device
            n = lv.tick_end()
            assert n == lv.picture_count
            collect(t)
        assert lv.tick_end() == n                                 # no tick in flight: the last count again
        lv.tick_begin()                                           # nothing pending: nothing in flight
        lv.write(ids[0], plans[0][0])
        lv.tick_begin()
        with pytest.raises(RuntimeError):
            lv.tick_begin()                                       # a second tick beside the first
        assert lv.tick_end() == 1
        assert sum(lv.stream_info(i).evictions for i in ids.values()) >= 1
    for s in plans:
        assert got[s] == want[s][join[s]:], s
    assert sum(len(x) for x in got[3]) > 0


@pytest.mark.parametrize("name", ["cfg0_240p_intra", "skipped_pictures_352x288", "uncovered_first_p_118x197", "odd_size_17x33", "long_gop_p_chain"])
def test_live_ragged_pieces_beside_ticks_in_flight(name, hip_lib):
    """bytes in arbitrary pieces written WHILE ticks that take only complete pictures are on the device: still the pictures of
    the whole stream decoded in one piece (the held tail, the cursor and the stamps are the tick's; the pieces wait their turn)"""
    fx, es, offs = load_case(os.path.join(ROOT, "tests", "golden", "frames_%s.json" % name))
    rng = random.Random(len(es))
    mean = max(64, len(es) // (3 * fx["n_frames"]))
    got, at = [], 0
    with jl.Live(fx["info"]["width"], fx["info"]["height"], 1, pictures_per_tick=3, store_bytes=2 * len(es) + 4096) as lv:
        s = lv.open()
        lv.write(s, es[:10]); at = 10
        while at < len(es):
            lv.tick_begin(flush=False)
            for _ in range(rng.choice([0, 1, 2, 4])):
                n = min(len(es) - at, rng.choice([1, 3, 17, mean // 2, mean, 2 * mean, 5 * mean]))
                lv.write(s, es[at:at + n])
                at += n
            lv.tick_end()
            drain(lv, got)
        for _ in range(fx["n_frames"]):
            lv.tick(flush=False)
            drain(lv, got)
        lv.tick(flush=True)
        drain(lv, got)
        assert lv.stream_info(s).pending_bytes == 0
    assert got == fx["frame_md5"]


def test_live_stream_of_another_size_is_refused_not_decoded(hip_lib):
    es, offs = synth.generate_config("cfg1_720p", n_frames=3, width=176, height=144)
    ok, offs_ok = synth.generate_config("cfg1_720p", n_frames=3, width=352, height=288)
    with jl.Live(352, 288, 2) as lv:
        a, b = lv.open(), lv.open()
        lv.write(a, es)
        lv.write(b, ok)
        assert lv.tick(flush=True) == 3
        assert {p.stream for p in lv.pictures()} == {b}
        ia = lv.stream_info(a)
        assert ia.status == 1 and ia.has_sequence_header and (ia.width, ia.height) == (176, 144) and ia.pending_bytes == 0
        lv.write(a, es)
        assert lv.tick(flush=True) == 0 and lv.stream_info(a).pending_bytes == 0


def test_live_bytes_before_the_first_header_are_skipped(hip_lib, libs):
    """a stream joined in the middle of a GOP: P pictures arrive before the first sequence header and are consumed
    undecoded, like the reference's write() does before it has a header (mpeg1.c:812-819)"""
    W, H = 352, 288
    es, offs = synth.generate_config("cfg1_720p", n_frames=30, stream=3, width=W, height=H)
    offs = [int(o) for o in offs]
    writes = picture_writes(es, offs)[5:]                    # from the sixth picture of the first GOP on
    want = oracle_fed(libs, [[w] for w in writes], 1 << 18)
    got = []
    with jl.Live(W, H, 1, store_bytes=1 << 18) as lv:
        s = lv.open()
        for w in writes:
            lv.write(s, w)
            lv.tick(flush=True)
            tick = []
            drain(lv, tick)
            got.append(tick)
    assert got == want and got[0] == [] and sum(len(t) for t in got) == 30 - 12


def test_live_read_frames_all_at_once(hip_lib):
    """jsmpeg_hip_live_read_frames: every picture of a tick in one call into pinned memory == the per-picture reads; ranges,
    a stride wider than the planes, refusals"""
    W, H = 352, 288
    gen = [synth.generate_config("cfg1_720p", n_frames=4, stream=70 + s, width=W, height=H) for s in range(5)]
    with jl.Live(W, H, 5, pictures_per_tick=3) as lv:
        ids = [lv.open() for _ in range(5)]
        for s, (es, offs) in enumerate(gen):
            for w in picture_writes(es, [int(o) for o in offs])[:1 + s % 3]:
                lv.write(ids[s], w)
        n = lv.tick(flush=True)
        assert n == sum(1 + s % 3 for s in range(5))
        one_by_one = [np.concatenate(lv.read_frame(i)) for i in range(n)]
        allf = lv.read_frames()
        assert allf.shape == (n, lv.luma_bytes + 2 * lv.chroma_bytes)
        assert all((allf[i] == one_by_one[i]).all() for i in range(n))
        part = lv.read_frames(2, 3).copy()
        assert all((part[k] == one_by_one[2 + k]).all() for k in range(3))
        assert lv.read_frames(n, 0).shape[0] == 0
        planes = lv.luma_bytes + 2 * lv.chroma_bytes
        wide = np.full((n, planes + 64), 0xEE, dtype=np.uint8)          # pageable memory, a stride with room behind each picture
        assert lv.L.jsmpeg_hip_live_read_frames(lv.h, 0, n, wide.ctypes.data, planes + 64) == 0
        assert all((wide[i, :planes] == one_by_one[i]).all() for i in range(n)) and (wide[:, planes:] == 0xEE).all()
        assert lv.L.jsmpeg_hip_host_register(wide.ctypes.data, wide.nbytes) == 0        # the caller's own memory, pinned
        wide[:] = 0
        assert lv.L.jsmpeg_hip_live_read_frames(lv.h, 0, n, wide.ctypes.data, planes + 64) == 0
        assert all((wide[i, :planes] == one_by_one[i]).all() for i in range(n))
        assert lv.L.jsmpeg_hip_host_unregister(wide.ctypes.data) == 0
        assert lv.L.jsmpeg_hip_live_read_frames(lv.h, 1, n, wide.ctypes.data, planes + 64) < 0     # past the last picture
        assert lv.L.jsmpeg_hip_live_read_frames(lv.h, 0, 1, wide.ctypes.data, planes - 1) < 0      # pictures would overlap


def test_live_rgba_and_device_frames(hip_lib, libs):
    """the RGBA stage on a live picture; device_frame is where the planes lie"""
    from oracle import checkers
    W, H = 352, 288
    es, offs = synth.generate_config("cfg1_720p", n_frames=4, width=W, height=H)
    frames, _, _ = cabi.decode_stream(libs["oracle"], es, keep="planes")
    with jl.Live(W, H, 1) as lv:
        s = lv.open()
        lv.write(s, es)
        assert lv.tick(flush=True) == 4
        pics = lv.pictures()
        assert len({p.device_frame for p in pics}) == 4 and all(p.device_frame for p in pics)
        for i in (0, 3):
            assert np.array_equal(lv.read_rgba(i), checkers.oracle_rgba(libs["oracle"], *frames[i], W, H))
        t = lv.timings()
        assert t["total_ms"] > 0 and t["parse_ms"] > 0 and t["recon_ms"] > 0


def test_live_streams_fed_as_transport_streams(hip_lib, libs):
    """jsmpeg_hip_live_write_ts: three streams as MPEG-TS bytes in ragged pieces (cut packets: the demuxer's leftover bytes),
    round-robin, a tick per round -- the library's ts.js restatement in front of every stream, its state kept between calls.
    Every picture == the oracle's decode of the stream, pts as the PES headers say, bytes as the reference demuxer delivers them."""
    from oracle import checkers
    W, H, N = 352, 288, 14
    tss, want, want_pts, want_bytes = [], [], [], []
    for s in range(3):
        es, offs = synth.generate_config("cfg1_720p", n_frames=N, stream=300 + s, width=W, height=H)
        ts = synth.mux_ts(es, offs)
        tss.append(ts)
        demuxed, writes = checkers.oracle_ts_demux(libs["oracle"], ts, 0xE0)
        want_pts.append([round(w[0], 6) for w in writes])
        want_bytes.append(sum(w[2] for w in writes))
        frames, _, _ = cabi.decode_stream(libs["oracle"], demuxed)
        want.append(frames)
    rng = random.Random(3)
    got, pts = {0: [], 1: [], 2: []}, {0: [], 1: [], 2: []}
    with jl.Live(W, H, 3, pictures_per_tick=4, store_bytes=1 << 18) as lv:
        ids = [lv.open() for _ in range(3)]
        at = [0, 0, 0]
        while any(at[s] < len(tss[s]) for s in range(3)):
            for s in range(3):
                n = min(len(tss[s]) - at[s], rng.choice([1, 187, 188, 189, 1000, 4000, 9000]))
                if n:
                    lv.write_ts(ids[s], tss[s][at[s]:at[s] + n])
                    at[s] += n
            lv.tick(flush=True)
            per = {}
            for p in drain(lv, None, per):
                pts[ids.index(p.stream)].append(round(p.pts, 6))
            for i, frames in per.items():
                got[ids.index(i)] += frames
        for s in range(3):
            assert lv.stream_info(ids[s]).bytes_written == want_bytes[s]
    for s in range(3):
        assert got[s] == want[s], s
        assert pts[s] == want_pts[s], s


def test_live_damaged_streams_do_not_disturb_their_neighbours(hip_lib, libs):
    """streams are independent: three streams fed noise, bit-flipped and truncated data beside three good ones, every tick --
    no fault, no hang, and the good streams' pictures are the oracle's, picture by picture"""
    W, H, N = 352, 288, 18
    rng = np.random.RandomState(21)
    good = []
    for s in range(3):
        es, offs = synth.generate_config("cfg1_720p", n_frames=N, stream=700 + s, width=W, height=H)
        good.append(picture_writes(es, [int(o) for o in offs]))
    want = [oracle_fed(libs, [[w] for w in good[s]], 1 << 18) for s in range(3)]
    es_b, offs_b = synth.generate_config("cfg1_720p", n_frames=N, stream=777, width=W, height=H)
    bad_src = picture_writes(es_b, [int(o) for o in offs_b])
    with jl.Live(W, H, 6, pictures_per_tick=4, store_bytes=1 << 18) as lv:
        g = [lv.open() for _ in range(3)]
        noise, flipped, cut = lv.open(), lv.open(), lv.open()
        for k in range(N):
            for s in range(3):
                lv.write(g[s], good[s][k])
            lv.write(noise, rng.randint(0, 256, size=rng.randint(1, 20000)).astype(np.uint8))
            w = bad_src[k].copy()
            if k:                                               # (the first write carries the sequence header: keep it, damage what follows)
                for _ in range(8):
                    w[rng.randint(0, len(w))] ^= 1 << rng.randint(0, 8)
            lv.write(flipped, w)
            lv.write(cut, bad_src[k][:max(1, len(bad_src[k]) - rng.randint(0, len(bad_src[k]) // 2))])
            lv.tick(flush=True)
            per = {}
            drain(lv, None, per)
            for s in range(3):
                assert per.get(g[s], []) == want[s][k], (k, s)
        assert lv.stream_info(noise).pictures == 0
        assert all(lv.stream_info(i).pending_bytes == 0 for i in (noise, flipped, cut))


@pytest.mark.parametrize("chunk", [None, "3000"], ids=["default", "staged_bytes_sent_every_3000"])
def test_live_fuzz_short_run(hip_lib, libs, chunk):
    """tools/fuzz_live.py: random sizes / syntax / feeding (whole pictures with small stores that evict, arbitrary byte pieces,
    TS in pieces), streams joining at random ticks -- a short run of the sweep whose long runs are profiles/r06_fuzz_live.txt.
    Second form: the staged bytes go to the device in chunks while the host is still writing (live.hip live_send_staged;
    1 MiB by default, more than these small pictures ever stage) -- with a chunk of 3000 bytes every case sends many, around
    evacuations and compactions of the staging buffer."""
    import subprocess
    import sys
    env = dict(os.environ)
    if chunk:
        env["JSMPEG_HIP_LIVE_UPLOAD_CHUNK"] = chunk
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_live.py"), "30", "77"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_read_out_beside_the_next_ticks(hip_lib, libs):
    """jsmpeg_hip_live_read_frames_begin / _end: a tick's pictures on their way to the host WHILE the next tick decodes into the
    same rings.  Four streams, rings of 3 + 2 frames, ticks of 1, 2 and 3 pictures per stream: with at most two pictures per
    stream in flight the next tick runs beside the copies (it writes the rings' other slots), with three it waits for them on the
    device -- either way what arrives is what the tick decoded.  Every fifth read-out stays in flight across TWO ticks (the second
    one waits for it on the device; their own pictures are read the plain way meanwhile).  A second _begin is refused, _end with
    nothing in flight is nothing."""
    from jsmpeg_amd import batch as jb
    n_streams, n_pic = 4, 36
    cases = [synth.generate_config("cfg1_720p", n_frames=n_pic, width=352, height=288, stream=120 + s) for s in range(n_streams)]
    want = [cabi.decode_stream(libs["oracle"], es)[0] for es, _ in cases]            # md5(Y | Cr | Cb) per picture
    writes = [picture_writes(es, [int(o) for o in offs]) for es, offs in cases]
    rng = random.Random(7)
    seen = [0] * n_streams

    def check(arr, streams_of):
        assert len(arr) == len(streams_of)
        for row, s in zip(arr, streams_of):
            assert hashlib.md5(row.tobytes()).hexdigest() == want[s][seen[s]], (s, seen[s])
            seen[s] += 1

    with jl.Live(352, 288, n_streams, pictures_per_tick=3, store_bytes=1 << 20) as lv:
        assert lv.read_frames_end() is None and lv.L.jsmpeg_hip_live_read_frames_end(lv.h) == 0
        for _ in range(n_streams):
            lv.open()
        at = [0] * n_streams
        inflight, saved, step, held, deep = None, [], 0, 0, 0
        while min(at) < n_pic:
            step += 1
            k = rng.choice([1, 1, 2, 2, 3])
            deep += k == 3
            for s in range(n_streams):
                for w in writes[s][at[s]:at[s] + k]:
                    lv.write(s, w)
                at[s] = min(n_pic, at[s] + k)
            lv.tick(flush=True)
            now = [p.stream for p in lv.pictures()]
            if inflight is not None and inflight["hold"]:
                inflight["hold"] -= 1
                saved.append((lv.read_frames().copy(), now))         # this tick's pictures the plain way; the read-out stays in flight
                continue
            if inflight is not None:
                check(lv.read_frames_end(), inflight["streams"])
                for a in saved:
                    check(*a)
                saved = []
            lv.read_frames_begin()
            assert lv.L.jsmpeg_hip_live_read_frames_begin(lv.h, 0, 0, None, 0) < 0 and "in flight" in jb.last_error()
            inflight = {"streams": now, "hold": 1 if step % 5 == 0 else 0}
            held += inflight["hold"]
        check(lv.read_frames_end(), inflight["streams"])
        for a in saved:
            check(*a)
        assert seen == [n_pic] * n_streams and held >= 2 and deep >= 2


@pytest.mark.parametrize("noise_tail", [b"\x00", b"\x00\x00", b"\x00\x00\x01"], ids=["00", "0000", "000001"])
def test_a_header_behind_held_bytes_survives_an_evacuation_before_any_tick(noise_tail, hip_lib, libs):
    """Noise that ENDS with what could begin a start code is held (a header may begin in it); the next write brings the sequence
    header and the first picture; a third write does not fit and evacuates the store -- all before any tick.  The reference found
    its header inside the second write() (mpeg1.c:812-819), so the stream goes on decoding what is written afterwards
    (tools/fuzz_live.py seed 43 case 377: the header went with the evacuated bytes, the stream never decoded anything).
    (Noise that ends with a sequence header's start code and a few bytes is outside this: the reference parses whatever its store
    holds behind the written bytes as the header's fields.)"""
    es, offs = synth.generate_config("cfg1_720p", n_frames=5, width=176, height=144, stream=77)
    ws = picture_writes(es, [int(o) for o in offs])
    noise = np.frombuffer(bytes(range(7, 250)) + noise_tail, np.uint8)
    store = len(noise) + len(ws[0]) + len(ws[1]) // 2          # the second picture does not fit beside the first
    writes = [noise, ws[0], ws[1], ws[2], ws[3], ws[4]]
    want = []
    with cabi.Mpeg1Decoder(libs["oracle"], store, cabi.MODE_EVICT) as dec:
        for w in writes[:3]:
            dec.write(w)
        while dec.decode():
            want.append(md5_planes(dec.planes()))
        for w in writes[3:]:
            dec.write(w)
            while dec.decode():
                want.append(md5_planes(dec.planes()))
    got = []
    with jl.Live(176, 144, 1, pictures_per_tick=4, store_bytes=store) as lv:
        s = lv.open()
        for w in writes[:3]:
            lv.write(s, w)
        lv.tick(flush=True)
        drain(lv, got)
        assert lv.stream_info(s).has_sequence_header and lv.stream_info(s).evictions >= 1
        for w in writes[3:]:
            lv.write(s, w)
            lv.tick(flush=True)
            drain(lv, got)
    assert len(want) >= 3 and got == want
