"""Parity tests proper of the MP2 audio stage: the HIP path (through the C ABI in libjsmpeg_hip.so) against the
committed golden fixtures and against the oracle on the same seeded inputs.  Bit-exact (binary32 PCM compared as
bit patterns): the stage follows the reference C's arithmetic operation by operation.  Needs an MI355X."""
import numpy as np
import pytest

from jsmpeg_amd import cabi, mp2, synth
from oracle import checkers
from mp2_util import FIXTURES, FIXTURE_IDS, frame_md5, load_case, same_bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", FIXTURES, ids=FIXTURE_IDS)
def test_batch_matches_golden(path, hip_lib):
    fx, data, offs = load_case(path)
    with mp2.Mp2Batch(1, len(data) + 64) as b:
        b.upload([data])
        assert b.decode() == fx["n_frames"]
        pcm = b.read_pcm(0)
        assert frame_md5(pcm) == fx["frame_md5"]
        for k in (0, fx["n_frames"] // 2, fx["n_frames"] - 1):
            off, size, rate = b.frame_info(0, k)
            assert off == int(offs[k]) and size == fx["frame_bytes"][k]
        assert b.frame_info(0, fx["n_frames"] - 1)[2] == fx["sample_rate"]
        # decoding the resident batch again gives the same samples (nothing is carried over between decodes)
        assert b.decode() == fx["n_frames"]
        assert same_bits(b.read_pcm(0), pcm)


@pytest.mark.parametrize("path", FIXTURES, ids=FIXTURE_IDS)
def test_decoder_abi_matches_golden(path, hip_lib):
    """The reference's 10-function ABI: one write, pull every frame."""
    fx, data, offs = load_case(path)
    pcm, idx, sizes, rate = cabi.decode_mp2_stream(hip_lib, data)
    assert frame_md5(pcm) == fx["frame_md5"]
    assert idx == fx["bit_index_after_decode"] and sizes == fx["frame_bytes"] and rate == fx["sample_rate"]


def test_decoder_abi_streaming_evict(hip_lib):
    """EVICT store smaller than the stream, a frame written / a frame pulled (how ts.js + Player drive it): the
    synthesis state lives on the device across calls and evictions."""
    fx, data, offs = load_case(FIXTURES[FIXTURE_IDS.index("varying_44k")])
    pcm, _, sizes, _ = cabi.decode_mp2_stream(hip_lib, data, offs, buffer_size=4096, mode=cabi.MODE_EVICT)
    assert frame_md5(pcm) == fx["frame_md5"] and sizes == fx["frame_bytes"]


def test_decoder_abi_expand_and_seek(hip_lib, libs):
    """Store that has to grow; then set_index back to an earlier frame: the reference keeps its synthesis state
    across a seek (mp2.c has no reset), so the re-decoded frames differ from the first pass exactly as the oracle's do."""
    fx, data, offs = load_case(FIXTURES[FIXTURE_IDS.index("stereo_44k_192")])
    out = {}
    for path in (hip_lib, libs["oracle"]):
        got = []
        with cabi.Mp2Decoder(path, 1024) as dec:
            dec.write(data)
            for _ in range(10):
                assert dec.decode() > 0
                got.append(np.stack(dec.channels()))
            dec.index = int(offs[3]) * 8
            while dec.decode():
                got.append(np.stack(dec.channels()))
        out[path] = np.stack(got)
    assert len(out[hip_lib]) == 10 + fx["n_frames"] - 3
    assert same_bits(out[hip_lib], out[libs["oracle"]])


def test_decoder_abi_refuses_what_the_reference_refuses(hip_lib, libs):
    _, data, offs = load_case(FIXTURES[FIXTURE_IDS.index("stereo_44k_192")])
    bad = data.copy()
    bad[int(offs[5]) + 1] = 0xF5          # Layer III
    pcm, idx, sizes, _ = cabi.decode_mp2_stream(hip_lib, bad)
    want = cabi.decode_mp2_stream(libs["oracle"], bad)
    assert len(pcm) == 5 and idx == want[1] and same_bits(pcm, want[0])
    with cabi.Mp2Decoder(hip_lib, 4096) as dec:
        assert dec.decode() == 0 and dec.sample_rate == 44100          # nothing buffered; mp2.c:234
        dec.write(np.array([0xFF], dtype=np.uint8))
        assert dec.decode() == 0                                       # fewer than 16 bits
        dec.write(np.array([0xFD, 0xF4, 0x00] + [0] * 100, dtype=np.uint8))
        assert dec.decode() == 0 and dec.index == 0                    # forbidden bit rate index


def test_batch_many_streams_vs_oracle(hip_lib, libs):
    """Streams of different lengths and configurations in one batch, an empty one, one whose last frame is cut,
    one without any frame; every sample against the oracle."""
    streams = [synth.generate_mp2_config(name, 9 + 11 * i, stream=60 + i)[0] for i, name in enumerate(synth.MP2_CONFIGS)]
    streams.insert(2, np.zeros(0, np.uint8))
    streams.append(streams[0][:len(streams[0]) - 100])
    streams.append(np.zeros(300, np.uint8))
    with mp2.Mp2Batch(len(streams), sum(len(s) for s in streams) + 64) as b:
        b.upload(streams)
        total = b.decode()
        assert total == b.frame_count() == sum(b.frame_count(s) for s in range(len(streams)))
        for s, data in enumerate(streams):
            want, _, sizes, _ = cabi.decode_mp2_stream(libs["oracle"], data)
            if len(want) and sum(sizes) > len(data):
                want = want[:-1]          # the one-frame ABI decodes a cut last frame (missing bytes read as 0), the batch does not
            assert b.frame_count(s) == len(want), s
            assert same_bits(b.read_pcm(s), want), "stream %d" % s
        assert b.frame_count(2) == 0 and b.frame_count(len(streams) - 1) == 0
        t = b.timings()
        assert t["total_ms"] > 0


def test_batch_full_size_vs_oracle(hip_lib, libs):
    """The audio that goes with the video benchmark batch: 64 stereo streams of 154 frames (4 s at 44.1 kHz);
    a sample of streams bit for bit against the oracle, all of them through a digest of the device buffer."""
    streams = [synth.generate_mp2_config("mp2_stereo_44k_192", 154, stream=s)[0] for s in range(64)]
    with mp2.Mp2Batch(64, sum(len(s) for s in streams) + 64) as b:
        b.upload(streams)
        assert b.decode() == 64 * 154
        for s in (0, 1, 31, 63):
            want = cabi.decode_mp2_stream(libs["oracle"], streams[s])[0]
            assert same_bits(b.read_pcm(s), want), "stream %d" % s
        # identical streams decode identically wherever they sit in the batch
        b.upload([streams[5]] * 3 + streams[:8])
        assert b.decode() == 11 * 154
        a = b.read_pcm(0)
        assert same_bits(a, b.read_pcm(1)) and same_bits(a, b.read_pcm(2)) and same_bits(a, b.read_pcm(3 + 5))


def _av_ts(video_case_frames, audio_fixture, seed_stream):
    """A TS with a video stream (0xE0), an audio stream (0xC0, PES_packet_length set, two frames per PES) and null
    packets, interleaved."""
    from ts_craft import Muxer
    es, voffs = synth.generate_config("cfg1_720p", n_frames=video_case_frames, width=176, height=144, stream=seed_stream)
    fx, data, aoffs = load_case(FIXTURES[FIXTURE_IDS.index(audio_fixture)])
    m = Muxer()
    a = 0
    for k in range(video_case_frames):
        end = len(es) if k == video_case_frames - 1 else int(voffs[k + 1])
        m.pes(0x100, 0xE0, es[int(voffs[k]):end].tobytes(), pts=90000 + 3000 * k)
        while a < fx["n_frames"] and a * 1152 / fx["sample_rate"] <= (k + 1) / 30.0:
            hi = min(a + 2, fx["n_frames"])
            m.pes(0x101, 0xC0, data[int(aoffs[a]):int(aoffs[hi])].tobytes(), pts=90000 + int(90000 * 1152 * a / fx["sample_rate"]),
                  with_length=True)
            a = hi
        m.packet(0x1fff, b"")
    while a < fx["n_frames"]:
        hi = min(a + 2, fx["n_frames"])
        m.pes(0x101, 0xC0, data[int(aoffs[a]):int(aoffs[hi])].tobytes(), pts=90000 + int(90000 * 1152 * a / fx["sample_rate"]),
              with_length=True)
        a = hi
    return m.bytes(), es, fx, data


def test_batch_ts_in_audio_and_video_from_the_same_buffers(hip_lib, libs):
    """The same MPEG-TS buffers handed to the video batch (stream 0xE0) and to the MP2 batch (stream 0xC0): the
    device demux delivers exactly what the ts.js restatement delivers, and both decode bit-exactly."""
    from jsmpeg_amd import batch as jb
    from jsmpeg_amd import hashing
    cases = [_av_ts(9, "stereo_44k_192", 3), _av_ts(6, "mono_32k_48", 4), _av_ts(12, "varying_44k", 5)]
    ts = [c[0] for c in cases]
    with mp2.Mp2Batch(len(ts), sum(len(t) for t in ts)) as ab, \
            jb.Batch(176, 144, len(ts), 40, sum(len(t) for t in ts) + 4096) as vb:
        ab.upload_ts(ts)                       # 0xC0
        vb.upload_ts(ts)                       # 0xE0
        assert ab.decode() == sum(c[2]["n_frames"] for c in cases)
        n_pics = vb.decode()
        assert n_pics == 9 + 6 + 12
        dev = vb.frame_hashes()
        per_stream = {}
        for p, info in enumerate(vb.pictures()):
            per_stream.setdefault(info.stream, []).append(int(dev[p]))
        for s, (tsb, es, fx, data) in enumerate(cases):
            want_bytes, want_writes = checkers.oracle_ts_demux(libs["oracle"], tsb, 0xC0)
            got_writes = ab.ts_writes(s)
            assert [(w[1], w[2]) for w in got_writes] == [(w[1], w[2]) for w in want_writes]
            assert all(abs(g[0] - w[0]) < 1e-9 for g, w in zip(got_writes, want_writes))
            delivered = sum(w[2] for w in want_writes)
            assert np.array_equal(ab.read_bytes(s), want_bytes[:delivered]) and delivered == len(data)
            assert frame_md5(ab.read_pcm(s)) == fx["frame_md5"]
            frames, _, _ = cabi.decode_stream(libs["oracle"], es, keep="planes")
            assert per_stream[s] == [hashing.frame_hash(*f) for f in frames]


def test_randomised_sweep(hip_lib, libs):
    """120 random generator configurations in ONE batch (every sampling frequency, bit rates / modes / CRC / padding
    changing from frame to frame, forbidden-but-decodable codes, sparse and dense allocations), every sample against
    the oracle; a dozen of them also through the one-frame ABI."""
    rng = np.random.RandomState(777)
    streams = []
    for case in range(120):
        kw = dict(sample_rate_index=int(rng.randint(0, 3)), bitrate_index=int(rng.randint(1, 15)), mode=int(rng.randint(0, 4)),
                  crc=int(rng.randint(0, 2)), vary=int(rng.rand() < 0.5), quirks=int(rng.rand() < 0.3),
                  alloc_permille=int(rng.choice([150, 500, 800, 1000])), sf_lo=int(rng.choice([8, 12, 30])), sf_hi=62,
                  seed=int(rng.randint(1, 2 ** 31 - 1)))
        streams.append(synth.generate_mp2(int(rng.randint(1, 12)), **kw)[0])
    want = [cabi.decode_mp2_stream(libs["oracle"], s)[0] for s in streams]
    with mp2.Mp2Batch(len(streams), sum(len(s) for s in streams) + 64) as b:
        b.upload(streams)
        assert b.decode() == sum(len(w) for w in want)
        bad = [i for i in range(len(streams)) if not same_bits(b.read_pcm(i), want[i])]
        assert not bad, bad
    for i in range(0, 120, 10):
        assert same_bits(cabi.decode_mp2_stream(hip_lib, streams[i])[0], want[i]), i


def test_damaged_streams_still_equal_the_oracle(hip_lib, libs):
    """Bit flips, random runs, truncation, dropped bytes -- 240 damaged streams in one batch: no fault, no hang, and
    the PCM still equals the oracle's bit for bit (the accumulator saturates where a damaged scalefactor drives it
    out of range, in the kernels and in the oracle alike); a sample of them also through the one-frame ABI."""
    from test_mp2_sim_device_functions import _damaged
    rng = np.random.RandomState(2025)
    streams, want, full = [], [], []
    for case in range(240):
        data, _ = synth.generate_mp2_config(list(synth.MP2_CONFIGS)[case % len(synth.MP2_CONFIGS)], 6, stream=900 + case)
        bad = _damaged(rng, data)
        pcm, idx, sizes, _ = cabi.decode_mp2_stream(libs["oracle"], bad)
        full.append((pcm, idx))
        if len(pcm) and sum(sizes) > len(bad):
            pcm = pcm[:-1]
        streams.append(bad)
        want.append(pcm)
    with mp2.Mp2Batch(len(streams), sum(len(s) for s in streams) + 64) as b:
        b.upload(streams)
        assert b.decode() == sum(len(w) for w in want)
        bad_ones = [i for i in range(len(streams)) if not same_bits(b.read_pcm(i), want[i])]
        assert not bad_ones, bad_ones
    for i in range(0, 240, 12):
        pcm, idx, _, _ = cabi.decode_mp2_stream(hip_lib, streams[i])
        assert idx == full[i][1] and same_bits(pcm, full[i][0]), i


def test_batch_upload_from_device_memory(hip_lib, libs):
    """Streams that are already in HBM (one packed buffer + byte ranges, the layout a rank holds after the RCCL
    scatter of its shard): same PCM as the host upload."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    streams = [synth.generate_mp2_config(name, 7 + 3 * i, stream=80 + i)[0] for i, name in enumerate(synth.MP2_CONFIGS)]
    packed, begin, end = [], [], []
    at = 0
    for s in streams:
        gap = np.full(5 + len(begin) % 3, 0xFF, np.uint8)      # ranges need not be aligned or adjacent
        packed.append(gap); at += len(gap)
        begin.append(at); packed.append(s); at += len(s); end.append(at)
    packed = np.concatenate(packed)
    dptr = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(len(packed))) == 0
    try:
        assert hip.hipMemcpy(dptr, ctypes.c_void_p(packed.ctypes.data), ctypes.c_size_t(len(packed)), 1) == 0
        with mp2.Mp2Batch(len(streams), len(packed) + 64) as b:
            b.upload_device(dptr, len(packed), begin, end)
            assert b.decode() == sum(7 + 3 * i for i in range(len(streams)))
            for i, s in enumerate(streams):
                assert same_bits(b.read_pcm(i), cabi.decode_mp2_stream(libs["oracle"], s)[0]), i
    finally:
        hip.hipFree(dptr)


def test_frame_that_promises_more_bits_than_it_has(hip_lib, libs):
    """An allocation that claims more sample bits than the frame's length holds: the reference reads on into the bytes
    behind the frame, so do the batch and the one-frame ABI (the kernels stage what a frame's fields can reach, not
    its length)."""
    from test_mp2_sim_device_functions import _overcommitted_stream
    streams = [_overcommitted_stream(seed) for seed in range(6)]
    want = [cabi.decode_mp2_stream(libs["oracle"], s, max_frames=1)[0] for s in streams]
    with mp2.Mp2Batch(len(streams), sum(len(s) for s in streams) + 64) as b:
        b.upload(streams)
        b.decode()
        for i in range(len(streams)):
            assert b.frame_count(i) >= 1 and same_bits(b.read_pcm(i, 0, 1), want[i]), i
    for i, s in enumerate(streams):
        assert same_bits(cabi.decode_mp2_stream(hip_lib, s, max_frames=1)[0], want[i]), i
