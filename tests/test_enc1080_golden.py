"""tests/golden/enc1080/: four 1080p GOPs from the test-side encoder (coded-video statistics at the headline's picture size and
bit rate; tools/enc_content.py makes them, make_golden_enc1080.py pins them: reference JS == wasm == C == the restatement).
CPU: the oracle against the golden vectors.  GPU: the batch path, the one-picture ABI and live streams against them."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from jsmpeg_amd import batch as jb
from jsmpeg_amd import build, cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "enc1080", "frames_enc1080_*.json")))
IDS = [os.path.basename(p)[7:-5] for p in CASES]
END = np.frombuffer(bytes([0, 0, 1, 0xB7]), np.uint8)


def load(path):
    fx = json.load(open(path))
    es = np.fromfile(os.path.join(os.path.dirname(path), fx["case"] + ".m1v"), dtype=np.uint8)
    assert hashlib.md5(es.tobytes()).hexdigest() == fx["es_md5"]
    return fx, es


def md5_planes(planes):
    h = hashlib.md5()
    for p in planes:
        h.update(p.tobytes())
    return h.hexdigest()


def test_there_are_four_cases():
    assert len(CASES) == 4


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_oracle_matches_golden(path, libs):
    fx, es = load(path)
    assert sorted(fx["agreed_by"]) == ["oracle", "ref_js", "ref_native", "ref_wasm"]
    frames = cabi.decode_stream(libs["oracle"], es)[0]
    assert frames == fx["frame_md5"]


@pytest.mark.gpu
def test_batch_of_all_gops_and_of_their_rotations_matches_golden(hip_lib):
    """the four GOPs as four streams, and two streams of all four GOPs one behind the other (later sequence headers are start codes
    the picture scan passes over, mpeg1.c:814): every picture == the golden vectors"""
    cases = [load(p) for p in CASES]
    singles = [es for _, es in cases]
    rot = [np.concatenate([cases[(k + r) % 4][1][:-4] for k in range(4)] + [END]) for r in (0, 3)]
    want = [fx["frame_md5"] for fx, _ in cases] + [sum((cases[(k + r) % 4][0]["frame_md5"] for k in range(4)), []) for r in (0, 3)]
    streams = singles + rot
    with jb.Batch(1920, 1080, len(streams), sum(len(w) for w in want) + 8, sum(len(s) for s in streams) + 64 * len(streams) + 4096) as b:
        b.upload(streams)
        n = b.decode()
        assert n == sum(len(w) for w in want)
        got = {}
        for p, i in enumerate(b.pictures()):
            assert i.decoded
            got.setdefault(i.stream, []).append(md5_planes(b.read_frame(p)))
        for s, w in enumerate(want):
            assert got[s] == w, s


@pytest.mark.gpu
@pytest.mark.parametrize("path", CASES[:2], ids=IDS[:2])
def test_one_picture_abi_and_live_stream_match_golden(path, hip_lib):
    """the reference's 15-function ABI fed picture by picture (ts.js's writes), and a live stream fed the same writes with a tick
    after each: the pictures of the golden vectors"""
    from jsmpeg_amd import live as jl
    fx, es = load(path)
    at = np.flatnonzero((es[:-3] == 0) & (es[1:-2] == 0) & (es[2:-1] == 1) & (es[3:] == 0))
    offs = np.concatenate([[0], at[1:], [len(es) - 4]]).astype(np.uint32)
    assert cabi.decode_stream(hip_lib, es, offs)[0] == fx["frame_md5"]
    got = []
    with jl.Live(1920, 1080, 1, pictures_per_tick=4, store_bytes=2 * 1024 * 1024) as l:
        s = l.open()
        for k in range(len(offs) - 1):
            end = len(es) if k == len(offs) - 2 else int(offs[k + 1])
            l.write(s, es[int(offs[k]):end], pts=k / 30.0)
            for i in range(l.tick(flush=True)):
                got.append(md5_planes(l.read_frame(i)))
    assert got == fx["frame_md5"]
