"""Diagnostics: phase timeline of k_recon workgroups (needs a -DJM_EXP_TIMING build and JSMPEG_HIP_TIMING=1).
Prints, per wavefront index of a workgroup, the mean cycles between the stamps of the last reconstruct launch."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["JSMPEG_HIP_TIMING"] = "1"
import bench
from jsmpeg_amd import batch as jb, synth
cfg = synth.CONFIGS[bench.CONFIG]
streams = [g[0] for g in bench.generate_streams(0, 64, 120)]
total = sum(len(s) for s in streams)
with jb.Batch(cfg["width"], cfg["height"], 64, 64 * 120 + 8, total + 64 * 64 + 4096) as b:
    b.upload(streams)
    b.decode(); b.decode()
    print(b.timings())
    L = b.L
    L.jsmpeg_hip_batch_debug_read.restype = ctypes.c_int
    L.jsmpeg_hip_batch_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
    n = 1900
    buf = np.zeros(n * 4 * 8, dtype=np.uint64)
    assert L.jsmpeg_hip_batch_debug_read(b.h, 8, buf.ctypes.data, 0, buf.nbytes) == 0
    t = buf.reshape(n, 4, 8).astype(np.int64)
    ok = (t[:, 0, 0] > 0) & (t[:, 0, 7] > 0)
    t = t[ok]
    print("workgroups sampled", len(t))
    names = ["desc+rec+front", "barrier1(+qm)", "konst+scatter+predict(wait pred)", "barrier2", "idct", "barrier3", "back+stores"]
    for w in range(4):
        d = np.diff(t[:, w, :], axis=1)
        print("wave %d:" % w, " ".join("%s=%.0f" % (nm, x) for nm, x in zip(names, d.mean(axis=0))), " total=%.0f" % (t[:, w, 7] - t[:, w, 0]).mean())
    print("workgroup lifetime (first stamp of any wave -> last): %.0f cycles" % (t[:, :, 7].max(axis=1) - t[:, :, 0].min(axis=1)).mean())
