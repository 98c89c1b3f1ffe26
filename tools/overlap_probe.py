"""Experiment: does running the slice parse of one half of the batch beside the reconstruct of the other half pay?
Two half-size batches on two HIP streams, decoded in loops by two host threads (the second starts a few ms late so the
phases interleave), against the same two batches decoded one after the other.
    python tools/overlap_probe.py [reps]"""
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from jsmpeg_amd import batch as jb, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hip = ctypes.CDLL("libamdhip64.so")
cfg = synth.CONFIGS[bench.CONFIG]
gen = bench.generate_streams(0, 64, 120)
streams = [g[0] for g in gen]
halves = [streams[:32], streams[32:]]
bs, sts = [], []
for h in halves:
    total = sum(len(s) for s in h)
    b = jb.Batch(cfg["width"], cfg["height"], 32, 32 * 120 + 8, total + 64 * 32 + 4096)
    b.upload(h)
    b.decode()
    bs.append(b)
    s = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(s)) == 0
    sts.append(s)

t0 = time.perf_counter()
for r in range(reps):
    for b in bs:
        b.decode()
seq = (time.perf_counter() - t0) / reps


def loop(i, delay):
    time.sleep(delay)
    for r in range(reps):
        bs[i].decode(stream=sts[i])


for delay in (0.0, 0.003, 0.005, 0.008):
    th = [threading.Thread(target=loop, args=(0, 0.0)), threading.Thread(target=loop, args=(1, delay))]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    par = (time.perf_counter() - t0 - delay * 0) / reps
    print("delay %.0f ms: sequential %.2f ms per 64 streams, two streams interleaved %.2f ms" % (delay * 1e3, seq * 1e3, par * 1e3))
