"""Parity tests proper of the LIVE audio streams (include/jsmpeg_hip.h part 6: streams that go on, one tick decodes the buffered
frames of all of them): the HIP path through the C ABI against the golden fixtures and against the oracle's decoder given
the SAME write() calls.  Bit-exact (binary32 PCM compared as bit patterns).  Needs an MI355X."""
import numpy as np
import pytest

from jsmpeg_amd import cabi, mp2, synth
from oracle import checkers
from mp2_util import FIXTURES, FIXTURE_IDS, frame_md5, load_case, same_bits

pytestmark = pytest.mark.gpu


def _bounds(data, offs):
    return [int(o) for o in offs] + [len(data)]


def test_every_fixture_frame_by_frame_in_one_handle(hip_lib):
    """All fixtures as streams of ONE handle, a frame written per stream and tick (one PES per frame): every tick decodes one
    frame of every stream that still has one; per stream the golden samples."""
    cases = [load_case(p) for p in FIXTURES]
    with mp2.Mp2Live(len(cases), max_frames_per_tick=2) as live:
        ids = [live.open() for _ in cases]
        assert ids == list(range(len(cases)))
        got = [[] for _ in cases]
        for k in range(max(c[0]["n_frames"] for c in cases)):
            fed = []
            for s, (fx, data, offs) in enumerate(cases):
                if k < fx["n_frames"]:
                    b = _bounds(data, offs)
                    live.write(s, 10.0 + k, data[b[k]:b[k + 1]])
                    fed.append(s)
            assert live.tick() == len(fed)
            frames = live.frames()
            assert [f["stream"] for f in frames] == fed
            pcm = live.read_pcm()
            for i, f in enumerate(frames):
                fx, data, offs = cases[f["stream"]]
                assert f["stream_offset"] == int(offs[k]) and f["bytes"] == fx["frame_bytes"][k] and f["pts"] == 10.0 + k
                got[f["stream"]].append(pcm[i])
        for s, (fx, data, offs) in enumerate(cases):
            assert frame_md5(got[s]) == fx["frame_md5"], FIXTURE_IDS[s]
            info = live.stream_info(s)
            assert info["frames"] == fx["n_frames"] and info["pending_bytes"] == 0 and info["sample_rate"] == fx["sample_rate"]
            assert info["bytes_written"] == len(data) and info["evictions"] == 0 and info["stalled"] == 0
        assert live.tick() == 0 and live.frames() == []


def test_whole_frame_writes_with_evictions_equal_the_oracle_given_the_same_writes(hip_lib, libs):
    """Writes of one to four whole frames (what ts.js hands over), ticks only now and then, a store of 4 KiB: writes find the
    store full of undecoded frames and throw them away (buffer.c:166-180).  The oracle's decoder (EVICT, the same capacity) gets
    the same writes and `while (decode())` at every tick: frame for frame the same samples, the same number of evictions."""
    rng = np.random.RandomState(4242)
    names = ["mp2_stereo_44k_192", "mp2_varying_44k", "mp2_mono_32k_48", "mp2_dual_44k_384", "mp2_joint_48k_128"]
    streams = [synth.generate_mp2_config(n, 60, stream=200 + i) for i, n in enumerate(names)]
    store = 4096
    decs = [cabi.Mp2Decoder(libs["oracle"], store, cabi.MODE_EVICT) for _ in streams]
    try:
        with mp2.Mp2Live(len(streams), max_frames_per_tick=20, store_bytes=store) as live:
            for _ in streams:
                live.open()
            at = [0] * len(streams)
            total = [0] * len(streams)
            for step in range(400):
                for s, (data, offs) in enumerate(streams):
                    b = _bounds(data, offs)
                    if at[s] >= len(offs) or rng.randint(4) == 0:
                        continue
                    k = min(int(rng.randint(1, 5)), len(offs) - at[s])
                    while b[at[s] + k] - b[at[s]] > store:
                        k -= 1
                    piece = data[b[at[s]]:b[at[s] + k]]
                    live.write(s, float(step), piece)
                    decs[s].write(piece)
                    at[s] += k
                if rng.randint(3):
                    continue
                n = live.tick()
                frames = live.frames()
                pcm = live.read_pcm()
                i = 0
                for s in range(len(streams)):
                    while True:
                        size = decs[s].decode()
                        if not size:
                            break
                        assert i < n and frames[i]["stream"] == s and frames[i]["bytes"] == size, (step, s, i)
                        assert same_bits(pcm[i], np.stack(decs[s].channels())), (step, s, i)
                        assert frames[i]["sample_rate"] == decs[s].sample_rate
                        i += 1
                        total[s] += 1
                assert i == n, (step, i, n)
            ev = [live.stream_info(s)["evictions"] for s in range(len(streams))]
            assert sum(ev) > 5 and min(total) > 10, (ev, total)
    finally:
        for d in decs:
            d.close()


def test_ragged_pieces_give_the_whole_streams_frames(hip_lib, libs):
    """Bytes in arbitrary pieces (frames cut anywhere, ticks that find nothing complete, ticks that find more frames than a
    tick takes): per stream the frames of the whole stream decoded in one piece."""
    rng = np.random.RandomState(99)
    streams = [synth.generate_mp2_config(name, 12 + 5 * i, stream=300 + i)[0] for i, name in enumerate(synth.MP2_CONFIGS)]
    want = [cabi.decode_mp2_stream(libs["oracle"], s)[0] for s in streams]
    n = len(streams)
    with mp2.Mp2Live(n, max_frames_per_tick=3) as live:
        for _ in range(n):
            live.open()
        at = [0] * n
        got = [[] for _ in range(n)]
        for _ in range(600):
            for s in range(n):
                if at[s] < len(streams[s]) and rng.randint(3):
                    k = int(rng.choice([1, 5, 97, 400, 1500, 5000]))
                    live.write(s, 0.0, streams[s][at[s]:at[s] + k])
                    at[s] += k
            cnt = live.tick()
            frames = live.frames()
            pcm = live.read_pcm()
            per = {}
            for i in range(cnt):
                got[frames[i]["stream"]].append(pcm[i])
                per[frames[i]["stream"]] = per.get(frames[i]["stream"], 0) + 1
            assert all(v <= 3 for v in per.values())
            if all(at[s] >= len(streams[s]) and live.stream_info(s)["pending_bytes"] == 0 for s in range(n)):
                break
        for s in range(n):
            assert len(got[s]) == len(want[s]) and same_bits(np.array(got[s]), want[s]), s


def test_streams_join_and_leave_and_ids_are_reused(hip_lib, libs):
    """A stream that takes over the id of one that left starts from a silent synthesis state (its ring is cleared), while the
    others go on undisturbed."""
    a, aoffs = synth.generate_mp2_config("mp2_stereo_44k_192", 10, stream=1)
    b, boffs = synth.generate_mp2_config("mp2_dual_44k_384", 10, stream=2)
    c, coffs = synth.generate_mp2_config("mp2_varying_44k", 8, stream=3)
    want = {k: cabi.decode_mp2_stream(libs["oracle"], v)[0] for k, v in (("a", a), ("b", b), ("c", c))}
    with mp2.Mp2Live(2, max_frames_per_tick=4) as live:
        sa, sb = live.open(), live.open()
        with pytest.raises(RuntimeError, match="in use"):
            live.open()
        ba, bb, bc = _bounds(a, aoffs), _bounds(b, boffs), _bounds(c, coffs)
        got = {"a": [], "b": [], "c": []}
        live.write(sa, 0, a[:ba[4]])
        live.write(sb, 0, b[:bb[3]])
        assert live.tick() == 7
        pcm = live.read_pcm()
        got["a"] += list(pcm[:4]); got["b"] += list(pcm[4:])
        live.close_stream(sa)
        with pytest.raises(RuntimeError, match="not open"):
            live.write(sa, 0, a[:10])
        sc = live.open()
        assert sc == sa
        live.write(sc, 0, c)
        live.write(sb, 0, b[bb[3]:])
        while live.tick():
            frames, pcm = live.frames(), live.read_pcm()
            for i, f in enumerate(frames):
                got["c" if f["stream"] == sc else "b"].append(pcm[i])
        assert same_bits(np.array(got["c"]), want["c"]) and same_bits(np.array(got["b"]), want["b"])
        assert same_bits(np.array(got["a"]), want["a"][:4])


def test_a_stalled_stream_recovers_by_evacuation_like_the_reference(hip_lib, libs):
    """Noise at the cursor: decode() returns 0 there for good (mp2.c:283-302), writes pile up behind it until one does not fit
    and evacuates the store (buffer.c:166-180) -- decoding resumes with that write.  The oracle's decoder gets the same writes."""
    data, offs = synth.generate_mp2_config("mp2_stereo_44k_192", 30, stream=9)
    b = _bounds(data, offs)
    store = 4096
    with cabi.Mp2Decoder(libs["oracle"], store, cabi.MODE_EVICT) as dec, mp2.Mp2Live(1, store_bytes=store) as live:
        s = live.open()
        writes = [data[b[0]:b[2]], np.frombuffer(b"\x55\xaa\x12\x34" * 40, np.uint8)] + [data[b[k]:b[k + 1]] for k in range(2, 30)]
        n_total = 0
        stalled_seen = False
        for w in writes:
            live.write(s, 0, w)
            dec.write(w)
            n = live.tick()
            frames, pcm = live.frames(), live.read_pcm()
            i = 0
            while True:
                size = dec.decode()
                if not size:
                    break
                assert i < n and frames[i]["bytes"] == size and same_bits(pcm[i], np.stack(dec.channels()))
                i += 1
            assert i == n
            n_total += n
            stalled_seen = stalled_seen or live.stream_info(s)["stalled"] == 1
        info = live.stream_info(s)
        assert stalled_seen and info["evictions"] >= 1 and info["stalled"] == 0 and n_total > 10


def test_limits_and_refusals(hip_lib):
    with mp2.Mp2Live(1, store_bytes=1000) as live:
        s = live.open()
        with pytest.raises(RuntimeError, match="larger than the stream's store"):
            live.write(s, 0, np.zeros(1001, np.uint8))
        live.write(s, 0, np.zeros(0, np.uint8))                  # an empty write is nothing
        live.write(s, 0, [np.zeros(400, np.uint8), np.zeros(600, np.uint8)])
        assert live.stream_info(s)["pending_bytes"] == 1000 and live.stream_info(s)["stalled"] == 1
        assert live.tick() == 0
        with pytest.raises(RuntimeError, match="not in the last tick"):
            live.read_pcm(0, 1)
        with pytest.raises(RuntimeError):
            live.stream_info(5)
    with pytest.raises(RuntimeError, match="too large"):
        mp2.Mp2Live(1 << 16, store_bytes=1 << 20)


def test_ts_in_audio_beside_the_live_video(hip_lib, libs):
    """The SAME MPEG-TS bytes, in ragged pieces, to a live video handle (stream id 0xE0) and a live audio handle (0xC0): the
    demuxer's state is kept per handle and stream; the audio frames are the fixture's, their time stamps the PES's."""
    from test_mp2_gpu import _av_ts
    from jsmpeg_amd import hashing
    from jsmpeg_amd import live as jl
    rng = np.random.RandomState(5)
    cases = [_av_ts(9, "stereo_44k_192", 3), _av_ts(6, "mono_32k_48", 4)]
    with mp2.Mp2Live(len(cases)) as al, jl.Live(176, 144, len(cases), pictures_per_tick=4) as vl:
        for _ in cases:
            assert al.open() == vl.open()
        at = [0] * len(cases)
        pcm = [[] for _ in cases]
        pts = [[] for _ in cases]
        pics = [[] for _ in cases]
        while any(at[s] < len(c[0]) for s, c in enumerate(cases)):
            for s, c in enumerate(cases):
                k = int(rng.choice([50, 188, 700, 3000]))
                piece = c[0][at[s]:at[s] + k]
                at[s] += k
                if len(piece):
                    al.write_ts(s, piece)
                    vl.write_ts(s, piece)
            n = al.tick()
            frames, got = al.frames(), al.read_pcm()
            for i in range(n):
                pcm[frames[i]["stream"]].append(got[i])
                pts[frames[i]["stream"]].append(frames[i]["pts"])
            vl.tick(flush=True)
            hs = vl.frame_hashes()
            for i, p in enumerate(vl.pictures()):
                pics[p.stream].append(int(hs[i]))
        for s, (tsb, es, fx, data) in enumerate(cases):
            assert frame_md5(pcm[s]) == fx["frame_md5"]
            _, want_writes = checkers.oracle_ts_demux(libs["oracle"], tsb, 0xC0)
            starts = {}
            for w in want_writes:                                   # (pts, offset, length) of every PES: a frame's pts is its PES's
                starts[w[1]] = w[0]
            off = 0
            for size, p in zip(fx["frame_bytes"], pts[s]):
                pes = max(o for o in starts if o <= off)
                assert abs(p - starts[pes]) < 1e-9
                off += size
            frames, _, _ = cabi.decode_stream(libs["oracle"], es, keep="planes")
            assert pics[s] == [hashing.frame_hash(*f) for f in frames]


def test_the_sub_block_count_near_its_ceiling(hip_lib, monkeypatch):
    """A stream that has been playing for 200 hours (JSMPEG_HIP_MP2_LIVE_N_ABS starts its count just below the step that keeps it
    under 2^31): the samples across the step are the golden ones"""
    fx, data, offs = load_case(FIXTURES[FIXTURE_IDS.index("varying_44k")])
    monkeypatch.setenv("JSMPEG_HIP_MP2_LIVE_N_ABS", str((1 << 30) + (1 << 29) - 36 * 4 - 16))
    b = _bounds(data, offs)
    got = []
    with mp2.Mp2Live(1, max_frames_per_tick=2) as live:
        s = live.open()
        for k in range(fx["n_frames"]):
            live.write(s, 0.0, data[b[k]:b[k + 1]])
            assert live.tick() == 1
            got.append(live.read_pcm()[0])
    assert frame_md5(got) == fx["frame_md5"]


def test_live_audio_fuzz_short_run(hip_lib, libs):
    """tools/fuzz_live_audio.py: random generator parameters / feeding (whole frames with small stores that evict and noise that
    stalls, arbitrary byte pieces, TS in pieces), streams joining and leaving on reused ids -- a short run of the sweep whose long
    runs are profiles/r06_fuzz_live_audio.txt"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_live_audio.py"), "60", "91"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
