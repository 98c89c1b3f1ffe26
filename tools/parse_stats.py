"""Turn statistics of k_parse (needs a -DJM_PARSE_STATS build and JSMPEG_HIP_DEBUG=4): per wavefront the turns, the
header steps run, and how many lanes each step kind found ready.   python tools/parse_stats.py [streams] [frames]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["JSMPEG_HIP_DEBUG"] = "4"
import numpy as np  # noqa: E402
import bench  # noqa: E402
bench.CONFIG = os.environ.get("JSMPEG_KBENCH_CONFIG", bench.CONFIG)
from jsmpeg_amd import batch as jb, synth  # noqa: E402

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 120
cfg = synth.CONFIGS[bench.CONFIG]
streams = [g[0] for g in bench.generate_streams(0, n_streams, frames)]
total = sum(len(s) for s in streams)
L = jb.lib()
L.jsmpeg_hip_batch_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
with jb.Batch(cfg["width"], cfg["height"], n_streams, n_streams * frames + 8, total + 64 * n_streams + 4096) as b:
    b.upload(streams)
    b.decode()
    n_slices = b.counters()["slices"]
    n_waves = max(1, n_slices // 4)   # 16 words per wavefront in a buffer of 4 words per start code (passes of 64 slices per wavefront)
    a = np.zeros((n_waves, 16), np.uint32)
    assert L.jsmpeg_hip_batch_debug_read(b.h, 8, a.ctypes.data, 0, a.nbytes) == 0, jb.last_error()
    a = a[a[:, 0] != 0xeeeeeeee]
    service = (a[:, 7] >> 20).astype(np.float64)
    a[:, 7] &= (1 << 20) - 1
    a = a.astype(np.float64)
    t = a[:, 0]
    print("wavefronts %d  turns per wavefront: mean %.0f  min %.0f  max %.0f" % (len(a), t.mean(), t.min(), t.max()))
    print("header steps per turn %.3f, ring services per turn %.3f" % (a[:, 1].sum() / t.sum(), service.sum() / t.sum()))
    for name, col in (("live", 7), ("blocked", 4), ("DC ready", 5), ("COEF ready (1st)", 2), ("SLOW ready", 6), ("COEF ready (2nd)", 3)):
        print("%-18s lanes per turn %.1f" % (name, a[:, col].sum() / t.sum()))
    ck = a[:, 8:15]
    if ck[:, 0].sum() > 0:      # shader clocks (s_memtime) of the turn loop and inside each of its parts, per wavefront
        tot = ck[:, 0].sum()
        print("shader clocks per turn %.0f; share of the turn loop: " % (tot / t.sum())
              + ", ".join("%s %.1f %%" % (n, 100.0 * ck[:, k].sum() / tot) for k, n in ((1, "ring service"), (2, "COLD"), (3, "DC"), (4, "COEF 1st"), (5, "SLOW"), (6, "COEF 2nd")))
              + ", rest (scheduling, waits between steps) %.1f %%" % (100.0 * (tot - ck[:, 1:7].sum()) / tot))
    print(b.timings())
