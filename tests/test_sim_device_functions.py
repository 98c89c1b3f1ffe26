"""The per-lane device functions (slice_parse.h, recon_block.h, index_tables.h),
compiled by g++ into a TEST-ONLY simulator (tests/sim/), against the golden
fixtures -- so their logic is checked in the build container, which has no
GPU.  The product never runs them on the CPU; the GPU-shaped parts (scan
compaction, LDS staging, launch order) are covered by the `-m gpu` tests."""
import ctypes
import glob
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from jsmpeg_amd import synth

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "frames_*.json")))


@pytest.fixture(scope="module")
def sim():
    so = os.path.join(ROOT, "tests", "sim", "libjsmpeg_sim.so")
    src = os.path.join(ROOT, "tests", "sim", "sim_decode.cpp")
    csrc = os.path.join(ROOT, "jsmpeg_amd", "csrc")
    deps = [src] + glob.glob(os.path.join(csrc, "*.h"))
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I", csrc,
                               "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.sim_decode_stream.restype = ctypes.c_int
    lib.sim_stale_windows.restype = ctypes.c_ulonglong
    lib.sim_decode_stream.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_int]
    return lib


@pytest.mark.parametrize("split", [1, 0], ids=["service_in_two_halves", "service_in_one_piece"])
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[7:-5] for p in FIXTURES])
def test_device_functions_match_golden(path, split, sim):
    sim.sim_split_service(split)       # jm_launch_parse picks the form per pass (kernels.hip): both are the product
    fx = json.load(open(path))
    es, _ = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
    fb = fx["info"]["coded_size"] * 3 // 2
    out = np.zeros(fx["n_frames"] * fb, dtype=np.uint8)
    n = sim.sim_decode_stream(es.ctypes.data, len(es), fx["info"]["width"], fx["info"]["height"], out.ctypes.data,
                              fx["n_frames"])
    assert n == fx["n_frames"]
    got = [hashlib.md5(out[i * fb:(i + 1) * fb].tobytes()).hexdigest() for i in range(n)]
    assert got == fx["frame_md5"]
    # the parse's carried bit window (slice_parse.h jm_win): every look found the bits the ring holds at its position
    assert sim.sim_stale_windows() == 0


def plan_reference(decoded, fwd, stream, covered, mb_size):
    """the rule in plain Python: a picture after its forward reference and, with unwritten macroblocks, after the decoded
    picture before last of its stream (what those macroblocks show, mpeg1.c:986-994)"""
    n = len(decoded)
    stale, level, last = [-1] * n, [0] * n, {}
    for p in range(n):
        if not decoded[p]:
            continue
        l1, l2 = last.get(stream[p], (-1, -1))
        stale[p] = l2
        last[stream[p]] = (p, l1)
        lv = level[fwd[p]] + 1 if fwd[p] >= 0 else 0
        if covered[p] < mb_size and stale[p] >= 0:
            lv = max(lv, level[stale[p]] + 1)
        level[p] = lv
    return stale, level


def run_plan(sim, decoded, fwd, stream, covered, mb_size, n_streams):
    n = len(decoded)
    arr = lambda v, t: np.asarray(v, dtype=t)
    d, f, s, c = arr(decoded, np.uint8), arr(fwd, np.int32), arr(stream, np.uint32), arr(covered, np.uint32)
    st, lv, unc = np.zeros(n, np.int32), np.zeros(n, np.int32), ctypes.c_uint32(0)
    sim.sim_plan.restype = ctypes.c_int
    sim.sim_plan.argtypes = [ctypes.c_uint32, ctypes.c_uint32] + [ctypes.c_void_p] * 4 + [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                                                                        ctypes.c_void_p]
    nl = sim.sim_plan(n, n_streams, d.ctypes.data, f.ctypes.data, s.ctypes.data, c.ctypes.data, mb_size, st.ctypes.data, lv.ctypes.data,
                      ctypes.byref(unc))
    return nl, list(st), list(lv), unc.value


def test_reconstruct_plan_hand_cases(sim):
    """I P P | I P P of one stream, 10 macroblocks per picture"""
    decoded, fwd, stream = [1] * 6, [-1, 0, 1, -1, 3, 4], [0] * 6
    full = [10] * 6
    # everything written: plain chains, two levels deep
    nl, st, lv, unc = run_plan(sim, decoded, fwd, stream, full, 10, 1)
    assert (nl, lv, unc) == (3, [0, 1, 2, 0, 1, 2], 0) and st == [-1, -1, 0, 1, 2, 3]
    # the first P of the second GOP leaves a macroblock unwritten: it shows picture 2 (level 2) -> level 3, its P after it
    cov = list(full); cov[4] = 9
    nl, st, lv, unc = run_plan(sim, decoded, fwd, stream, cov, 10, 1)
    assert (nl, lv, unc) == (5, [0, 1, 2, 0, 3, 4], 1)
    # the I picture of the second GOP does: after picture 1 (level 1) -> level 2, and its chain behind it
    cov = list(full); cov[3] = 0
    nl, st, lv, unc = run_plan(sim, decoded, fwd, stream, cov, 10, 1)
    assert (nl, lv, unc) == (5, [0, 1, 2, 2, 3, 4], 1)
    # unwritten macroblocks in the first two pictures of a stream show zeros: no dependency
    cov = list(full); cov[0] = 3; cov[1] = 3
    nl, st, lv, unc = run_plan(sim, decoded, fwd, stream, cov, 10, 1)
    assert (nl, lv, unc) == (3, [0, 1, 2, 0, 1, 2], 2)
    # a picture that is not decoded (B) neither counts nor rotates
    decoded2, fwd2 = [1, 0, 1, 1], [-1, -1, 0, 2]
    nl, st, lv, unc = run_plan(sim, decoded2, fwd2, [0] * 4, [10, 0, 10, 9], 10, 1)
    assert (nl, lv, st, unc) == (3, [0, 0, 1, 2], [-1, -1, -1, 0], 1)


def test_reconstruct_plan_random(sim):
    rng = np.random.default_rng(5)
    for case in range(200):
        n_streams = int(rng.integers(1, 5))
        n = int(rng.integers(1, 120))
        stream = sorted(int(x) for x in rng.integers(0, n_streams, size=n))          # pictures come stream after stream
        decoded = [int(rng.random() < 0.85) for _ in range(n)]
        fwd, last = [], {}
        for p in range(n):
            is_p = decoded[p] and stream[p] in last and rng.random() < 0.8
            fwd.append(last[stream[p]] if is_p else -1)
            if decoded[p]:
                last[stream[p]] = p
        covered = [int(10 if rng.random() < 0.7 else rng.integers(0, 10)) for _ in range(n)]
        st_ref, lv_ref = plan_reference(decoded, fwd, stream, covered, 10)
        nl, st, lv, unc = run_plan(sim, decoded, fwd, stream, covered, 10, n_streams)
        assert st == st_ref and lv == lv_ref
        assert nl == (max(lv_ref[p] for p in range(n) if decoded[p]) + 1 if any(decoded) else 0)
        assert unc == sum(1 for p in range(n) if decoded[p] and covered[p] < 10)


def run_plan_ordered(sim, decoded, fwd, stream, n_streams, group, slack=8):
    n = len(decoded)
    cap = 8 * (n + 8)
    seq = (ctypes.c_int32 * cap)()
    lock = ctypes.c_uint32(0)
    rows = sim.sim_plan_ordered(n, n_streams, (ctypes.c_uint8 * n)(*decoded), (ctypes.c_int32 * n)(*fwd), (ctypes.c_uint32 * n)(*stream),
                                group, slack, seq, cap, ctypes.byref(lock))
    prev, last = [-1] * n, {}
    for p in range(n):
        if decoded[p]:
            prev[p] = last.get(stream[p], -1)
            last[stream[p]] = p
    return rows, [seq[i] for i in range(8 * rows)], prev, lock.value


def test_ordered_plan_properties(sim):
    """The one-launch plan (recon_plan.h): every decoded picture once; a class = whole streams; a picture's predecessor in
    its stream comes earlier in the SAME class, `group` places back while the class still has that many streams going;
    no class more than the slack above the mean; batches that cannot fill eight classes are refused."""
    rng = np.random.default_rng(11)
    for case in range(150):
        n_streams = int(rng.integers(1, 40))
        per = [int(rng.integers(1, 30)) if rng.random() < 0.3 else 24 for _ in range(n_streams)]
        decoded, fwd, stream, last = [], [], [], {}
        for s_, n_ in enumerate(per):
            for i in range(n_):
                p = len(decoded)
                d = int(rng.random() < 0.9)
                is_p = d and s_ in last and i % 6 != 0
                decoded.append(d); stream.append(s_); fwd.append(last[s_] if is_p else -1)
                if d:
                    last[s_] = p
        group = int(rng.integers(1, 5))
        rows, seq, prev, lockstep = run_plan_ordered(sim, decoded, fwd, stream, n_streams, group)
        n_dec = sum(decoded)
        loads = {}
        for s_ in range(n_streams):
            loads[s_] = sum(1 for p in range(len(decoded)) if decoded[p] and stream[p] == s_)
        # the deal, restated: longest stream first onto the class with the least so far
        cls = [0] * 8
        for n_ in sorted((v for v in loads.values() if v), reverse=True):
            cls[cls.index(min(cls))] += n_
        fits = n_streams >= 8 and n_dec > 0 and max(cls) * 8 * 100 <= n_dec * 108
        assert (rows != 0) == fits
        if rows == 0:
            continue
        assert n_streams >= 8
        pos = {}
        for i, p in enumerate(seq):
            if p >= 0:
                assert p not in pos and decoded[p]
                pos[p] = i
        assert len(pos) == n_dec
        cls_of_stream = {}
        for p, i in pos.items():
            assert cls_of_stream.setdefault(stream[p], i % 8) == i % 8          # whole streams per class
        for c in range(8):
            col = [p for p in seq[c::8]]
            assert all(x < 0 for x in col[len([x for x in col if x >= 0]):])   # padding only at the end
        for p, i in pos.items():
            q = prev[p]          # the picture before it in its stream: what its forward reference / stale frame can be at the latest
            if q >= 0:
                assert pos[q] % 8 == i % 8 and pos[q] < i
                # every picture of its stream in between the two is impossible; the distance is the streams in lockstep
                # (a class walks `group` at a time, or a few more where that keeps its last set from being a small remainder)
                assert (i - pos[q]) // 8 <= 2 * group
        cls_load = [sum(1 for x in seq[c::8] if x >= 0) for c in range(8)]
        assert max(cls_load) == rows and max(cls_load) * 8 * 100 <= n_dec * 108
        assert lockstep == min(group, min(len({stream[p] for p in seq[c::8] if p >= 0}) for c in range(8)))


def test_ordered_plan_benchmark_shape(sim):
    """64 equal streams x 120 pictures, GOP 12, two streams in lockstep per class: 960 rows, no padding, predecessor exactly
    two places back except where a class moves on to its next pair of streams"""
    decoded, fwd, stream = [], [], []
    for s_ in range(64):
        for i in range(120):
            p = len(decoded)
            decoded.append(1); stream.append(s_); fwd.append(-1 if i % 12 == 0 else p - 1)
    rows, seq, prev, lockstep = run_plan_ordered(sim, decoded, fwd, stream, 64, 2)
    assert rows == 960 and all(p >= 0 for p in seq) and lockstep == 2
    pos = {p: i for i, p in enumerate(seq)}
    for p in range(len(decoded)):
        if prev[p] >= 0:
            assert (pos[p] - pos[prev[p]]) // 8 == 2


def test_gop_chains(sim):
    """jm_plan_chains: a chain begins at every decoded picture without a forward reference and at a stream's first decoded
    picture; pictures that are not decoded belong to none; a P picture is in its forward reference's chain"""
    rng = np.random.default_rng(23)
    for case in range(100):
        n_streams = int(rng.integers(1, 5))
        n = int(rng.integers(1, 90))
        stream = sorted(int(x) for x in rng.integers(0, n_streams, size=n))
        decoded = [int(rng.random() < 0.9) for _ in range(n)]
        fwd, last = [], {}
        for p in range(n):
            is_p = decoded[p] and stream[p] in last and rng.random() < 0.8
            fwd.append(last[stream[p]] if is_p else -1)
            if decoded[p]:
                last[stream[p]] = p
        out = (ctypes.c_uint32 * max(1, n))()
        k = sim.sim_plan_chains(n, n_streams, (ctypes.c_uint8 * n)(*decoded), (ctypes.c_int32 * n)(*fwd), (ctypes.c_uint32 * n)(*stream), out)
        chain = [out[p] for p in range(n)]
        heads = 0
        for p in range(n):
            if not decoded[p]:
                assert chain[p] == 0xFFFFFFFF
                continue
            if fwd[p] >= 0:
                assert chain[p] == chain[fwd[p]]
            else:
                heads += 1
                assert chain[p] == heads - 1                 # numbered in picture order
        assert k == heads
