"""Shared helpers of the MP2 tests."""
import ctypes
import glob
import hashlib
import json
import os
import subprocess

import numpy as np

from conftest import ROOT
from jsmpeg_amd import synth

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mp2_*.json")))
FIXTURE_IDS = [os.path.basename(p)[4:-5] for p in FIXTURES]


def load_case(path):
    fx = json.load(open(path))
    data, offs = synth.generate_mp2_config(fx["config"], fx["n_frames"], **fx["overrides"])
    assert hashlib.md5(data.tobytes()).hexdigest() == fx["stream_md5"], "generator drifted from the fixture"
    return fx, data, offs


def frame_md5(pcm):
    return [hashlib.md5(np.ascontiguousarray(f, dtype="<f4").tobytes()).hexdigest() for f in pcm]


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


_sim = None


def sim_lib():
    """tests/sim/sim_mp2.cpp: the device functions of the MP2 kernels compiled by g++ (TEST ONLY)."""
    global _sim
    if _sim is None:
        so = os.path.join(ROOT, "tests", "sim", "libjsmpeg_sim_mp2.so")
        src = os.path.join(ROOT, "tests", "sim", "sim_mp2.cpp")
        csrc = os.path.join(ROOT, "jsmpeg_amd", "csrc")
        deps = [src] + glob.glob(os.path.join(csrc, "mp2_*.h"))
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                                   "-I", csrc, "-o", so, src])
        _sim = ctypes.CDLL(so)
        _sim.sim_mp2_batch.restype = ctypes.c_int
    return _sim


def sim_batch(streams):
    lib = sim_lib()
    streams = [np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
    n = len(streams)
    ptrs = (ctypes.c_void_p * n)(*[s.ctypes.data for s in streams])
    lens = (ctypes.c_uint64 * n)(*[len(s) for s in streams])
    cap = sum(len(s) // 96 + 1 for s in streams)
    pcm = np.zeros((cap, 2, 1152), np.float32)
    ff = np.zeros(n + 1, np.uint32)
    r = lib.sim_mp2_batch(ptrs, lens, n, ctypes.c_void_p(pcm.ctypes.data), cap, ctypes.c_void_p(ff.ctypes.data))
    assert r >= 0
    return [pcm[ff[i]:ff[i + 1]] for i in range(n)]


class SimLive:
    """The host side of jsmpeg_hip_mp2_live_* restated in a few lines (stores, cursors, sub-block counts) around the
    simulator's sim_mp2_live_tick -- the kernels' LIVE placement (per-stream rings, frame places) on the CPU.  TEST ONLY."""

    def __init__(self, n_streams, cap, n_abs0=0):
        self.lib = sim_lib()
        self.lib.sim_mp2_live_tick.restype = None
        self.n, self.cap = n_streams, cap
        self.ring = 64
        while self.ring < 15 + 36 * cap:
            self.ring *= 2
        self.rings = np.zeros((n_streams, self.ring, 64), np.float32)
        self.n_abs = np.full(n_streams, n_abs0, np.uint32)
        self.store = [bytearray() for _ in range(n_streams)]

    def write(self, s, data):
        self.store[s] += bytes(data)

    def tick(self):
        """-> per stream float32[count, 2, 1152]"""
        bufs = [np.frombuffer(bytes(b), np.uint8) for b in self.store]
        ptrs = (ctypes.c_void_p * self.n)(*[b.ctypes.data if len(b) else None for b in bufs])
        lens = np.array([len(b) for b in bufs], np.uint32)
        pcm = np.zeros((self.n * self.cap, 2, 1152), np.float32)
        count = np.zeros(self.n, np.uint32)
        used = np.zeros(self.n, np.uint32)
        self.lib.sim_mp2_live_tick(ptrs, ctypes.c_void_p(lens.ctypes.data), self.n, self.cap, self.ring, ctypes.c_void_p(self.rings.ctypes.data),
                                   ctypes.c_void_p(self.n_abs.ctypes.data), ctypes.c_void_p(pcm.ctypes.data),
                                   ctypes.c_void_p(count.ctypes.data), ctypes.c_void_p(used.ctypes.data))
        out = []
        at = 0
        for s in range(self.n):
            assert count[s] <= self.cap
            out.append(pcm[at:at + count[s]].copy())          # packed in tick order
            at += int(count[s])
            del self.store[s][:int(used[s])]
            self.n_abs[s] += 36 * count[s]
            if self.n_abs[s] >= (1 << 30) + (1 << 29):            # mp2_live.hip: the count stays below 2^31 by steps every ring size divides
                self.n_abs[s] -= 1 << 29
        return out
