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
ALIGNED = [p for p in FIXTURES if "resync" not in p]
ALIGNED_IDS = [os.path.basename(p)[3:-5] for p in ALIGNED]


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
    es, writes = checkers.oracle_ts_demux(libs["oracle"], ts, fx["stream_id"])
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
@pytest.mark.parametrize("path", ALIGNED, ids=ALIGNED_IDS)
def test_device_demux_matches_reference_fixture(path, hip_lib):
    from jsmpeg_amd import batch as jb
    fx, ts = load_case(path)
    with jb.Batch(176, 144, 2, 64, 1 << 20) as b:
        b.upload_ts([ts, ts[:188 * 7]], fx["stream_id"])          # a second, shorter stream beside it
        es = b.read_es(0)
        assert as_fixture_writes(es, b.ts_writes(0)) == fx["writes"]
        assert hashlib.md5(es.tobytes()).hexdigest() == fx["total_md5"]


@pytest.mark.gpu
def test_device_demux_rejects_unaligned_input(hip_lib):
    from jsmpeg_amd import batch as jb
    ts = ts_craft.case_garbage_prefix_resync()
    with jb.Batch(176, 144, 1, 16, 1 << 20) as b:
        with pytest.raises(RuntimeError, match="sync byte"):
            b.upload_ts([ts])


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
