"""Randomised parity sweep on the CPU: the device functions of the slice parse and the reconstruct (slice_parse.h,
recon_block.h), compiled for the host by tests/sim, against the oracle (checker) on random picture sizes and generator
parameters -- what a change of the parse's steps can be checked with before a GPU is spent on it.  Also counts looks whose
carried bit window was not the ring's (must be 0).    python tools/fuzz_sim.py [cases] [seed]"""
import ctypes
import glob
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import build, cabi, synth  # noqa: E402

so = os.path.join(ROOT, "tests", "sim", "libjsmpeg_sim.so")
src = os.path.join(ROOT, "tests", "sim", "sim_decode.cpp")
csrc = os.path.join(ROOT, "jsmpeg_amd", "csrc")
if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in [src] + glob.glob(os.path.join(csrc, "*.h"))):
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I", csrc, "-o", so, src])
sim = ctypes.CDLL(so)
sim.sim_decode_stream.restype = ctypes.c_int
sim.sim_decode_stream.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
sim.sim_stale_windows.restype = ctypes.c_ulonglong

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    w, h = int(rng.integers(1, 40)) * 16 - int(rng.integers(0, 16)), int(rng.integers(1, 24)) * 16 - int(rng.integers(0, 16))
    ov = dict(width=max(w, 2), height=max(h, 2), gop=int(rng.choice([1, 2, 3, 5, 9, 12, 15, 40])), ac_max=int(rng.choice([0, 1, 3, 8, 24, 63])),
              qscale_lo=int(rng.integers(1, 8)), qscale_hi=int(rng.integers(8, 32)), escape_permille=int(rng.choice([0, 20, 300, 1000])),
              custom_quant=int(rng.integers(0, 2)), quirk_levels=int(rng.integers(0, 2)), dc_size_max=int(rng.integers(0, 9)),
              coded_permille=int(rng.choice([50, 400, 950])), f_code_max=int(rng.integers(1, 8)), syntax_quirks=int(rng.choice([0, 1, 4, 5])),
              mv_jitter=int(rng.choice([0, 0, 1, 2, 6])))
    n = int(rng.integers(2, 14))
    try:
        es, _ = synth.generate_config("cfg1_720p", n_frames=n, stream=7000 * c, **ov)
    except RuntimeError as e:
        print("case %d: generator: %s" % (c, e)); continue
    frames, _, info = cabi.decode_stream(build.LIB_ORACLE, es, keep="planes")
    fb = info["coded_size"] * 3 // 2 if isinstance(info, dict) and "coded_size" in info else None
    cw, ch = (ov["width"] + 15) // 16 * 16, (ov["height"] + 15) // 16 * 16
    fb = cw * ch * 3 // 2
    out = np.zeros((n + 2) * fb, dtype=np.uint8)
    sim.sim_split_service(c & 1)         # the ring service in two halves / in one piece (jm_launch_parse picks per pass)
    got = sim.sim_decode_stream(es.ctypes.data, len(es), ov["width"], ov["height"], out.ctypes.data, n + 2)
    ok = got == len(frames)
    if ok:
        for i, f in enumerate(frames):
            y, cr, cb = (np.ascontiguousarray(p).reshape(-1) for p in f)
            o = out[i * fb:(i + 1) * fb]
            if not (np.array_equal(o[:cw * ch], y) and np.array_equal(o[cw * ch:cw * ch * 5 // 4], cr) and np.array_equal(o[cw * ch * 5 // 4:], cb)):
                ok = False
                break
    if not ok:
        bad += 1
        print("case %d MISMATCH: got %d pictures, oracle %d; %r frames=%d" % (c, got, len(frames), ov, n), flush=True)
stale = sim.sim_stale_windows()
print("%d cases, %d mismatches, %d stale windows" % (cases, bad, stale))
sys.exit(1 if bad or stale else 0)
