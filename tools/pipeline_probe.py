"""Experiment: two FULL batches (the same 64 x 120 pictures each, own token / record / frame buffers) on two HIP streams,
decoded alternately by two host threads so that the slice parse of one pass runs beside the reconstruct of the pass before
it -- against one batch decoded pass after pass.     python tools/pipeline_probe.py [reps]"""
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: the decode library binds to the HIP runtime torch loads)
import bench  # noqa: E402
from jsmpeg_amd import batch as jb, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = synth.CONFIGS[bench.CONFIG]
gen = bench.generate_streams(0, 64, 120)
streams = [g[0] for g in gen]
total = sum(len(s) for s in streams)
bs, sts = [], []
N_BATCHES = int(os.environ.get("JSMPEG_PROBE_BATCHES", "2"))     # batches in flight (each its own HIP stream and host thread)
for k in range(N_BATCHES):
    b = jb.Batch(cfg["width"], cfg["height"], 64, 64 * 120 + 8, total + 64 * 64 + 4096)
    b.upload(streams)
    b.decode()
    b.decode()
    bs.append(b)
    keep = torch.cuda.Stream()
    sts.append((ctypes.c_void_p(keep.cuda_stream), keep))

t0 = time.perf_counter()
for r in range(reps):
    bs[0].decode()
seq = (time.perf_counter() - t0) / reps
print("one batch, pass after pass: %.2f ms per pass" % (seq * 1e3))


def loop(i, delay):
    time.sleep(delay)
    for r in range(reps):
        bs[i].decode(stream=sts[i][0])


for delay in (0.0, 0.004, 0.007, 0.010):
    th = [threading.Thread(target=loop, args=(i, delay * i)) for i in range(N_BATCHES)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    par = (time.perf_counter() - t0) / (N_BATCHES * reps)
    print("%d batches on %d streams, each started %.0f ms after the one before: %.2f ms per pass" % (N_BATCHES, N_BATCHES, delay * 1e3, par * 1e3))
