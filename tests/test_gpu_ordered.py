"""The ORDERED reconstruct (one launch per batch: every eighth of the GPU walks its streams in lockstep, a picture's tiles
wait for the picture before it in its stream; csrc/recon_plan.h jm_plan_ordered, kernels.hip k_recon) against the golden
fixtures and the oracle: same frames as one launch per dependency level, for every lockstep width, with streams of
unequal length, with pictures whose unwritten macroblocks show the decoded picture before last across a GOP boundary --
and the launch that flags itself is done over level by level before anybody reads a frame.  Needs an MI355X."""
import glob
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from jsmpeg_amd import batch as jb
from jsmpeg_amd import cabi, hashing, synth

pytestmark = pytest.mark.gpu

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "frames_*.json")))


def md5_planes(planes):
    h = hashlib.md5()
    for p in planes:
        h.update(p.tobytes())
    return h.hexdigest()


class order_env:
    """JSMPEG_HIP_RECON_ORDER is read when a batch is created"""
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("group", [1, 2, 3])
def test_ordered_launch_matches_golden(group, hip_lib):
    """every fixture of moderate size as 16 streams of one batch (two per class), every frame against the golden md5"""
    seen = 0
    for path in FIXTURES:
        fx = json.load(open(path))
        if fx["n_frames"] * fx["info"]["coded_size"] > 12e6 or "abi_frame_md5" in fx:
            continue
        es, _ = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
        n_streams = 16
        with order_env(JSMPEG_HIP_RECON_ORDER=group):
            b = jb.Batch(fx["info"]["width"], fx["info"]["height"], n_streams, n_streams * fx["n_frames"] + 4, n_streams * (len(es) + 64) + 8192)
        with b:
            b.upload([es] * n_streams)
            for rep in range(2):          # the second pass: the counts are reset, the epoch moves on
                assert b.decode() == n_streams * fx["n_frames"]
                info = b.recon_info()
                predicted = any(i.forward >= 0 for i in b.pictures())     # (a batch of intra pictures only has nothing to order: one plain launch)
                assert info["launches"] == 1 and info["group"] == (min(group, 2) if predicted else 0) and info["status"] == 0, (path, info)
                for p in range(n_streams * fx["n_frames"]):
                    if rep == 0 or p % 7 == 0:
                        assert md5_planes(b.read_frame(p)) == fx["frame_md5"][p % fx["n_frames"]], (os.path.basename(path), p)
        seen += 1
    assert seen >= 10


def test_ordered_equals_level_by_level_on_ragged_streams(hip_lib, libs):
    """24 streams with different seeds and different lengths (cut at picture boundaries), two GOP lengths: the ordered
    launch, the per-level launches and the oracle agree on every frame"""
    streams, want = [], []
    for s in range(24):
        es, offs = synth.generate_config("cfg1_720p", n_frames=26, stream=s, width=176, height=144, gop=(6 if s % 2 else 13))
        n = 26 - s % 3
        cut = es[:int(offs[n])] if n < 26 else es
        streams.append(cut)
        frames, _, _ = cabi.decode_stream(libs["oracle"], cut, keep="planes")
        want.append([hashing.frame_hash(*f) for f in frames])
    total = sum(len(w) for w in want)
    got = {}
    for order in (2, 0):
        with order_env(JSMPEG_HIP_RECON_ORDER=order):
            b = jb.Batch(176, 144, 24, total + 8, sum(len(s) for s in streams) + 8192)
        with b:
            b.upload(streams)
            assert b.decode() == total
            dev = b.frame_hashes()
            info = b.recon_info()
            assert (info["launches"] == 1) == (order != 0) and info["status"] == 0
            per = {}
            for p, pic in enumerate(b.pictures()):
                per.setdefault(pic.stream, []).append(int(dev[p]))
            got[order] = per
    for s in range(24):
        assert got[2][s] == want[s] and got[0][s] == want[s], "stream %d" % s


def test_batches_that_do_not_fill_eight_classes_go_level_by_level(hip_lib):
    es, _ = synth.generate_config("cfg1_720p", n_frames=13, stream=1, width=176, height=144)
    for n_streams, ordered in ((1, False), (7, False), (8, True), (9, False), (15, True), (16, True)):
        # 9 streams: one class carries two, 2 / (9 / 8) is far over the 8 % slack; 15: seven classes of two and one of one, 2 / (15 / 8) = 1.067
        with order_env(JSMPEG_HIP_RECON_ORDER=2):         # (forced: by itself a batch of pictures this small never takes the ordered launch)
            b = jb.Batch(176, 144, n_streams, n_streams * 13 + 4, n_streams * (len(es) + 64) + 8192)
        with b:
            b.upload([es] * n_streams)
            assert b.decode() == n_streams * 13
            assert (b.recon_info()["launches"] == 1) == ordered, n_streams


def test_the_lockstep_width_follows_the_picture_size(hip_lib):
    """by itself the engine walks as many streams in lockstep as put the last tile of a picture's forward reference ~400
    workgroups behind the picture's first tile in its class's dispatch order (as many as the class has, if fewer; at least 160
    workgroups), and launches level by level where the batch cannot give that (small pictures, few streams per class): 1080p x 16
    streams -> the class's two in lockstep (204 tiles each), one launch; 176x144 x 16 -> per level"""
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "frames_enc_static_1920x1080.json")))
    es, _ = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
    # (the fixture's intra picture is DENSE -- more than 19.4 bytes per macroblock: left to itself such a batch goes level by level
    #  so that its intra pictures get k_recon_intra_dense; JSMPEG_HIP_RECON_DENSE=0, read when a batch is created, takes that rule out)
    for dense_rule, want in ((0, dict(launches=1, group=2)), (None, dict(launches=fx["n_frames"], group=0))):
        with order_env(**({"JSMPEG_HIP_RECON_DENSE": dense_rule} if dense_rule is not None else {})):
            b = jb.Batch(1920, 1080, 16, 16 * fx["n_frames"] + 4, 16 * (len(es) + 64) + 8192)
        with b:
            b.upload([es] * 16)
            assert b.decode() == 16 * fx["n_frames"]
            info = b.recon_info()
            assert info["launches"] == want["launches"] and info["group"] == want["group"] and info["status"] == 0, (dense_rule, info)
            for p in range(16 * fx["n_frames"]):
                assert md5_planes(b.read_frame(p)) == fx["frame_md5"][p % fx["n_frames"]], p
    es, _ = synth.generate_config("cfg1_720p", n_frames=13, stream=1, width=176, height=144)
    with jb.Batch(176, 144, 16, 16 * 13 + 4, 16 * (len(es) + 64) + 8192) as b:
        b.upload([es] * 16)
        assert b.decode() == 16 * 13
        assert b.recon_info()["launches"] == 12


def test_a_launch_that_flags_itself_is_done_over(hip_lib):
    """JSMPEG_HIP_RECON_BREAK: one picture of the plan never reports, its successor's wait runs out of (shortened)
    patience, the launch flags itself -- the frames are rebuilt level by level before the sync returns, the result is
    the golden one, and the batch stays with per-level launches"""
    code = r'''
import hashlib, json, os, sys
sys.path.insert(0, %r)
from jsmpeg_amd import batch as jb, synth
fx = json.load(open(os.path.join(%r, "tests", "golden", "frames_long_gop_p_chain.json")))
es, _ = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
with jb.Batch(fx["info"]["width"], fx["info"]["height"], 8, 8 * fx["n_frames"] + 4, 8 * (len(es) + 64) + 8192) as b:
    b.upload([es] * 8)
    for rep in range(2):
        assert b.decode() == 8 * fx["n_frames"]
        info = b.recon_info()
        print("INFO", rep, info)
        assert (info["status"] != 0) == (rep == 0) and info["launches"] > 1
        for p in range(8 * fx["n_frames"]):
            h = hashlib.md5()
            for plane in b.read_frame(p):
                h.update(plane.tobytes())
            assert h.hexdigest() == fx["frame_md5"][p %% fx["n_frames"]], p
print("DONE")
''' % (ROOT, ROOT)
    env = dict(os.environ, JSMPEG_HIP_RECON_BREAK="8", JSMPEG_HIP_RECON_PATIENCE="2000", JSMPEG_HIP_RECON_ORDER="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "flagged itself" in r.stderr


def test_narrow_batches_walk_gop_chains(hip_lib, libs):
    """ONE stream of many GOPs (a file): the classes of the ordered launch walk GOP chains instead of streams -- one launch,
    every picture against the oracle"""
    es, _ = synth.generate_config("cfg1_720p", n_frames=192, stream=2, gop=6)
    frames, _, _ = cabi.decode_stream(libs["oracle"], es, keep="planes")
    want = [hashing.frame_hash(*f) for f in frames]
    with jb.Batch(1280, 720, 1, 200, len(es) + 8192) as b:
        b.upload([es])
        for rep in range(2):
            assert b.decode() == 192
            info = b.recon_info()
            assert info["launches"] == 1 and info["group"] >= 3 and info["status"] == 0, info
            assert [int(h) for h in b.frame_hashes()] == want


def test_a_chain_whose_first_pictures_leave_macroblocks_unwritten_is_done_over(hip_lib):
    """the assumption of the chain plan, broken on purpose: the fixture's first P pictures leave macroblocks unwritten that show
    the GOP before (another chain: possibly another class).  Forced onto GOP chains (JSMPEG_HIP_RECON_CHAINS), the decode
    notices once the parse's counts are in (status 4), reconstructs level by level, and the pictures are the golden ones."""
    code = r'''
import hashlib, json, os, sys
sys.path.insert(0, %r)
from jsmpeg_amd import batch as jb, synth
fx = json.load(open(os.path.join(%r, "tests", "golden", "frames_uncovered_first_p_118x197.json")))
es, _ = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
n = 6
with jb.Batch(fx["info"]["width"], fx["info"]["height"], n, n * fx["n_frames"] + 4, n * (len(es) + 64) + 8192) as b:
    b.upload([es] * n)
    for rep in range(2):
        assert b.decode() == n * fx["n_frames"]
        info = b.recon_info()
        print("INFO", info)
        assert info["status"] == 4 and info["launches"] > 1, info
        for p in range(n * fx["n_frames"]):
            h = hashlib.md5()
            for plane in b.read_frame(p):
                h.update(plane.tobytes())
            assert h.hexdigest() == fx["frame_md5"][p %% fx["n_frames"]], p
print("DONE")
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, JSMPEG_HIP_RECON_CHAINS="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_the_plan_is_a_setting_of_the_batch(hip_lib):
    """jsmpeg_hip_batch_set_reconstruct: 0 = one launch per dependency level, 1 = the engine's choice again; the pictures are the
    same either way (what a host with two batches in flight sets on wide batches)"""
    streams = [synth.generate_config("cfg2_1080p", n_frames=13, stream=s)[0] for s in range(16)]
    with jb.Batch(1920, 1080, 16, 16 * 13 + 4, sum(len(s) for s in streams) + 16 * 64 + 8192) as b:
        b.upload(streams)
        assert b.decode() == 16 * 13
        first, auto = b.frame_hashes().copy(), b.recon_info()
        assert auto["launches"] == 1 and auto["status"] == 0, auto
        b.set_reconstruct("levels")
        assert b.decode() == 16 * 13
        info = b.recon_info()
        assert info["launches"] > 1 and info["group"] == 0, info
        assert np.array_equal(b.frame_hashes(), first)
        b.set_reconstruct("auto")
        assert b.decode() == 16 * 13
        assert b.recon_info()["launches"] == 1 and np.array_equal(b.frame_hashes(), first)
        with pytest.raises(RuntimeError):
            b.set_reconstruct(7)
