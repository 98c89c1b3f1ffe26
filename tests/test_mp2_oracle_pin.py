"""Pins the MP2 oracle (oracle/mp2_oracle.c): against the committed golden fixtures everywhere, against the
reference's own C (oracle/_ref) where it was built, and against live runs of the reference's wasm and JS decoders
under Node where /root/reference exists.  Bit-exact for the C / wasm class, 2e-6 for mp2.js (see the oracle's
header for why the reference's two implementations differ)."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT, have_reference
from jsmpeg_amd import cabi, synth
from mp2_util import FIXTURES, FIXTURE_IDS, frame_md5, load_case, same_bits


@pytest.mark.parametrize("path", FIXTURES, ids=FIXTURE_IDS)
def test_oracle_matches_golden(path, libs):
    fx, data, offs = load_case(path)
    pcm, idx, sizes, rate = cabi.decode_mp2_stream(libs["oracle"], data)
    assert frame_md5(pcm) == fx["frame_md5"]
    assert idx == fx["bit_index_after_decode"] and sizes == fx["frame_bytes"] and rate == fx["sample_rate"]
    # frame by frame, the way ts.js hands over PES payloads (one write per frame, decode what is there)
    pcm2, _, sizes2, _ = cabi.decode_mp2_stream(libs["oracle"], data, offs)
    assert same_bits(pcm, pcm2) and sizes2 == sizes


@pytest.mark.parametrize("path", FIXTURES, ids=FIXTURE_IDS)
def test_reference_native_matches_golden(path, libs):
    if not libs["ref"] or not os.path.exists(libs["ref"]):
        pytest.skip("oracle/_ref not built (needs /root/reference once)")
    fx, data, offs = load_case(path)
    pcm, idx, sizes, rate = cabi.decode_mp2_stream(libs["ref"], data)
    assert frame_md5(pcm) == fx["frame_md5"]
    assert idx == fx["bit_index_after_decode"] and sizes == fx["frame_bytes"] and rate == fx["sample_rate"]


def test_oracle_evict_mode_streaming(libs):
    """EVICT (streaming) store smaller than the stream: write a frame, decode a frame (buffer.c:167-190); the
    synthesis state (V ring, v_pos) must carry across evictions."""
    fx, data, offs = load_case(FIXTURES[FIXTURE_IDS.index("varying_44k")])
    for lib in [libs["oracle"]] + ([libs["ref"]] if libs["ref"] and os.path.exists(libs["ref"]) else []):
        pcm, _, sizes, _ = cabi.decode_mp2_stream(lib, data, offs, buffer_size=4096, mode=cabi.MODE_EVICT)
        assert frame_md5(pcm) == fx["frame_md5"], lib


def test_oracle_stops_at_an_invalid_header(libs):
    """decode() returns 0 and leaves the cursor at a header that is not MPEG-1 Layer II (mp2.c:283-302); bytes
    written later do not help (the reference never resynchronises)."""
    _, data, offs = load_case(FIXTURES[FIXTURE_IDS.index("stereo_44k_192")])
    bad = data.copy()
    bad[int(offs[5]) + 1] = 0xF5          # layer bits -> Layer III
    for lib in [libs["oracle"]] + ([libs["ref"]] if libs["ref"] and os.path.exists(libs["ref"]) else []):
        pcm, idx, sizes, _ = cabi.decode_mp2_stream(lib, bad)
        assert len(pcm) == 5 and idx[-1] == int(offs[5]) * 8
        for hdr in ([0xFF, 0xFD, 0x04, 0x00], [0xFF, 0xFD, 0xF4, 0x00], [0xFF, 0xFD, 0x9C, 0x00], [0xFF, 0xF5, 0x94, 0x00]):
            # free format, forbidden bit rate, reserved sampling frequency, MPEG-2 ID: all refused
            with cabi.Mp2Decoder(lib, 4096) as dec:
                dec.write(np.array(hdr + [0] * 600, dtype=np.uint8))
                if hdr[2] == 0x04 and lib == libs["ref"]:
                    continue          # free format indexes the reference's bit rate table at -1: outside the contract
                assert dec.decode() == 0 and dec.index == 0


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference")
@pytest.mark.parametrize("impl", ["wasm", "js"])
def test_reference_under_node_matches_golden(impl, libs):
    fx, data, offs = load_case(FIXTURES[FIXTURE_IDS.index("varying_32k_quirks")])
    want = cabi.decode_mp2_stream(libs["oracle"], data)[0]
    with tempfile.TemporaryDirectory() as d:
        data.tofile(os.path.join(d, "a.mp2"))
        meta = json.loads(subprocess.check_output(["node", os.path.join(ROOT, "oracle", "ref_node_mp2.js"),
                                                   os.path.join(d, "a.mp2"), impl, os.path.join(d, "o.f32")]))
        got = np.fromfile(os.path.join(d, "o.f32"), dtype="<f4").reshape(-1, 2, 1152)
    assert meta["frames"] == fx["n_frames"] and meta["sampleRate"] == fx["sample_rate"]
    if impl == "wasm":
        assert frame_md5(got) == fx["frame_md5"]                       # bit-exact class
    else:
        assert float(np.abs(got.astype(np.float64) - want).max()) < 2e-6   # binary64 intermediates in mp2.js


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference")
def test_reference_ts_path_feeds_whole_frames(libs):
    """Audio through the reference's own demuxer (src/ts.js, stream 0xC0) into its wasm decoder, several frames
    per PES: the same PCM as the raw stream."""
    from ts_craft import Muxer
    fx, data, offs = load_case(FIXTURES[FIXTURE_IDS.index("joint_48k_128")])
    m = Muxer()
    for k in range(0, fx["n_frames"], 3):
        hi = min(k + 3, fx["n_frames"])
        m.pes(0x101, 0xC0, data[int(offs[k]):int(offs[hi])].tobytes(), pts=90000 + 2160 * k, with_length=True)
    with tempfile.TemporaryDirectory() as d:
        m.bytes().tofile(os.path.join(d, "a.ts"))
        meta = json.loads(subprocess.check_output(["node", os.path.join(ROOT, "oracle", "ref_node_mp2.js"),
                                                   os.path.join(d, "a.ts"), "wasm", os.path.join(d, "o.f32"), "--ts"]))
        got = np.fromfile(os.path.join(d, "o.f32"), dtype="<f4").reshape(-1, 2, 1152)
    assert meta["frames"] == fx["n_frames"] and meta["writes"] == (fx["n_frames"] + 2) // 3
    assert frame_md5(got) == fx["frame_md5"]


def test_randomised_sweep_restatement_equals_reference_c(libs):
    """300 random generator configurations: the restatement against the reference's own C, bit for bit (PCM, cursor,
    frame sizes, sampling rate)."""
    if not libs["ref"] or not os.path.exists(libs["ref"]):
        pytest.skip("oracle/_ref not built (needs /root/reference once)")
    rng = np.random.RandomState(4242)
    for case in range(300):
        kw = dict(sample_rate_index=int(rng.randint(0, 3)), bitrate_index=int(rng.randint(1, 15)), mode=int(rng.randint(0, 4)),
                  crc=int(rng.randint(0, 2)), vary=int(rng.rand() < 0.5), quirks=int(rng.rand() < 0.3),
                  alloc_permille=int(rng.choice([150, 500, 800, 1000])), sf_lo=int(rng.choice([8, 12, 30])), sf_hi=62,
                  seed=int(rng.randint(1, 2 ** 31 - 1)))
        data, _ = synth.generate_mp2(int(rng.randint(1, 7)), **kw)
        a = cabi.decode_mp2_stream(libs["oracle"], data)
        r = cabi.decode_mp2_stream(libs["ref"], data)
        assert same_bits(a[0], r[0]) and a[1:] == r[1:], (case, kw)
