"""On the GPU box: jsmpeg_hip_batch_upload_device's placement (k_place) on cfg2-sized input -- 64 packed streams of ~7.7 MB at
arbitrary offsets (15 of 16 not congruent to their 16-byte aligned destinations) and the same streams packed at 16-byte
boundaries; ms per call (the call waits for its stream) and GB/s moved (read + write).    python tools/place_bench.py"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from jsmpeg_amd import batch as jb  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
lens = [int(x) for x in rng.integers(7_600_000, 7_900_000, 64)]
for name, align in (("packed at any offset", 1), ("packed at 16-byte boundaries", 16)):
    begin, end, off = [], [], 0
    for n in lens:
        off = (off + align - 1) // align * align
        begin.append(off)
        off += n
        end.append(off)
    total = off
    d = torch.randint(0, 256, (total,), dtype=torch.uint8, device=dev)
    with jb.Batch(1920, 1088, 64, 64, total + 64 * 64 + 4096) as b:
        bg, en = np.asarray(begin, np.uint32), np.asarray(end, np.uint32)
        for _ in range(3):
            b.upload_device(ctypes.c_void_p(d.data_ptr()), total, bg, en)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            b.upload_device(ctypes.c_void_p(d.data_ptr()), total, bg, en)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        s = 17
        ok = np.array_equal(np.frombuffer(b.read_es(s), np.uint8), d[begin[s]:end[s]].cpu().numpy())
        print("%-30s %7.3f ms per upload_device of %d MB (fill + placement + tables; %.0f GB/s read + write)  stream %d %s"
              % (name, ms, total // 1000000, 2 * total / ms / 1e6, s, "ok" if ok else "DIFFERS"))
