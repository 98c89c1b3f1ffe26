"""Where a LIVE tick's slice parse spends its time (round 6): the pass of a tick -- 64 streams, ONE P picture each: 4352
slices, a handful per wavefront, every wavefront (nearly) alone on its SIMD -- with the turn statistics of a -DJM_PARSE_STATS
build (variants/stats.so), per lanes-per-wavefront setting.   JSMPEG_HIP_LIB=$PWD/variants/stats.so python tools/r06_tick_parse_stats.py
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["JSMPEG_HIP_DEBUG"] = "4"
import numpy as np  # noqa: E402
from jsmpeg_amd import batch as jb, synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # picture of the GOP: 0 = the I picture, 1.. = P pictures
streams = []
for s in range(S):
    es, offs = synth.generate_config("cfg2_1080p", n_frames=which + 2, stream=s)
    offs = [int(o) for o in offs]
    first = bytes(es[:offs[1]])
    head = first[:first.index(b"\x00\x00\x01\x00")]            # sequence (+ GOP) header in front of the first picture
    streams.append(np.frombuffer(head + bytes(es[offs[which]:offs[which + 1]]), dtype=np.uint8) if which else es[:offs[1]])
total = sum(len(x) for x in streams)
L = jb.lib()
L.jsmpeg_hip_batch_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
with jb.Batch(1920, 1080, S, S + 8, total + 64 * S + 4096) as b:
    b.upload(streams)
    for _ in range(3):
        b.decode()
    n_slices = b.counters()["slices"]
    a = np.zeros((max(1, n_slices), 16), np.uint32)
    assert L.jsmpeg_hip_batch_debug_read(b.h, 8, a.ctypes.data, 0, a.nbytes) == 0, jb.last_error()
    a = a[a[:, 0] != 0xeeeeeeee]
    service = (a[:, 7] >> 20).astype(np.float64)
    a[:, 7] &= (1 << 20) - 1
    a = a.astype(np.float64)
    t = a[:, 0]
    print("%d streams, picture %d of the GOP: %d slices, %d bytes per slice; wavefronts %d, turns per wavefront: mean %.0f max %.0f" % (S, which, n_slices, total // max(1, n_slices), len(a), t.mean(), t.max()))
    print("header steps per turn %.3f, ring services per turn %.3f" % (a[:, 1].sum() / t.sum(), service.sum() / t.sum()))
    for name, col in (("live", 7), ("blocked", 4), ("DC ready", 5), ("COEF ready (1st)", 2), ("SLOW ready", 6), ("COEF ready (2nd)", 3)):
        print("  %-18s lanes per turn %.2f" % (name, a[:, col].sum() / t.sum()))
    ck = a[:, 8:15]
    if ck[:, 0].sum() > 0:
        tot = ck[:, 0].sum()
        print("shader clocks per turn %.0f (the longest wavefront: %.0f turns x %.0f clocks = %.0f clocks); share of the turn loop: " % (tot / t.sum(), t.max(), ck[t.argmax(), 0] / t.max(), ck[t.argmax(), 0])
              + ", ".join("%s %.1f %%" % (n, 100.0 * ck[:, k].sum() / tot) for k, n in ((1, "ring service"), (2, "COLD"), (3, "DC"), (4, "COEF 1st"), (5, "SLOW"), (6, "COEF 2nd")))
              + ", rest (scheduling, waits between steps) %.1f %%" % (100.0 * (tot - ck[:, 1:7].sum()) / tot))
    print(b.timings())
