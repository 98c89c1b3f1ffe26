"""Turn-structure experiments on the CPU: the slice parse's wavefront scheduler as tests/sim emulates it (64 lanes, the same
lane functions, the same blocking rules), built with other step orders per turn, on one stream -- turns taken and a static
instruction estimate (per executed step kind, from the ISA of the product kernel).  What a change of the turn is worth in
TURNS (the walk of the longest slices) and in INSTRUCTIONS (the issue port) before a kernel is built for it.
    python tools/sim_turn_orders.py [intra|cfg2|cfg4]"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import synth  # noqa: E402

CSRC = os.path.join(ROOT, "jsmpeg_amd", "csrc")
SRC = os.path.join(ROOT, "tests", "sim", "sim_decode.cpp")
K = dict(COLD="JM_ST_COLD", DC="JM_ST_DC", COEF="JM_ST_COEF", SLOW="JM_ST_SLOW")
COST = (175, 42, 36, 59, 74, 0, 14)      # COLD, DC, COEF, SLOW, ring service, -, per turn (vector instructions, product ISA)

VARIANTS = [
    ("product: COLD DC COEF SLOW COEF", "COLD DC COEF SLOW COEF", 2, 0),
    ("+ DC COEF behind", "COLD DC COEF SLOW COEF DC COEF", 3, 1),
    ("+ DC behind", "COLD DC COEF SLOW COEF DC", 2, 1),
    ("+ COEF behind", "COLD DC COEF SLOW COEF COEF", 3, 0),
    ("COEF first: COLD COEF DC COEF SLOW COEF", "COLD COEF DC COEF SLOW COEF", 3, 0),
    ("two rounds: COLD DC COEF SLOW COEF DC COEF SLOW COEF", "COLD DC COEF SLOW COEF DC COEF SLOW COEF", 4, 1),
]


def build(tag, order, coef_repeat, extra_dc):
    so = "/tmp/sim_order_%s.so" % tag
    defs = ["-DJM_SIM_ORDER=" + ",".join(K[k] for k in order.split()), "-DJM_COEF_REPEAT=%d" % coef_repeat, "-DJM_EXTRA_DC=%d" % extra_dc]
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I", CSRC, "-o", so, SRC] + defs)
    lib = ctypes.CDLL(so)
    lib.sim_decode_stream.restype = ctypes.c_int
    lib.sim_decode_stream.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    lib.sim_picks.restype = ctypes.c_uint64
    lib.sim_cost.restype = ctypes.c_uint64
    return lib


what = sys.argv[1] if len(sys.argv) > 1 else "intra"
if what == "intra":      # cfg2's intra pictures alone: the wavefronts whose walk is cfg2's pass
    es, _ = synth.generate_config("cfg2_1080p", n_frames=2, gop=1)
    w, h, n = 1920, 1080, 2
elif what == "cfg4":
    es, _ = synth.generate_config("cfg4_2160p", n_frames=2, gop=1)
    w, h, n = 3840, 2160, 2
else:
    es, _ = synth.generate_config("cfg2_1080p", n_frames=13)
    w, h, n = 1920, 1080, 13
cw, ch = (w + 15) // 16 * 16, (h + 15) // 16 * 16
out = np.zeros((n + 1) * cw * ch * 3 // 2, dtype=np.uint8)
ref = None
print("%s: %d bytes, %d pictures" % (what, len(es), n))
for i, (name, order, rep, xdc) in enumerate(VARIANTS):
    lib = build(str(i), order, rep, xdc)
    lib.sim_reset_counters()
    lib.sim_kcost((ctypes.c_int * 7)(*COST))
    got = lib.sim_decode_stream(es.ctypes.data, len(es), w, h, out.ctypes.data, n + 1)
    digest = hash(out.tobytes())
    ref = digest if ref is None else ref
    picks, cost = lib.sim_picks(), lib.sim_cost()
    print("%-58s pictures %d %s  turns %8d  instructions %11d  per turn %5.1f" % (name, got, "same pictures" if digest == ref else "DIFFERENT PICTURES", picks, cost, cost / max(1, picks)))
