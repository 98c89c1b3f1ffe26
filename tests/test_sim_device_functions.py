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
    lib.sim_decode_stream.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_int]
    return lib


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[7:-5] for p in FIXTURES])
def test_device_functions_match_golden(path, sim):
    fx = json.load(open(path))
    es, _ = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
    fb = fx["info"]["coded_size"] * 3 // 2
    out = np.zeros(fx["n_frames"] * fb, dtype=np.uint8)
    n = sim.sim_decode_stream(es.ctypes.data, len(es), fx["info"]["width"], fx["info"]["height"], out.ctypes.data,
                              fx["n_frames"])
    assert n == fx["n_frames"]
    got = [hashlib.md5(out[i * fb:(i + 1) * fb].tobytes()).hexdigest() for i in range(n)]
    assert got == fx["frame_md5"]
