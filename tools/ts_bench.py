"""Ingest-side throughput: cfg2's 64 x 120-picture streams as MPEG-TS -> jsmpeg_hip_batch_upload_ts (device demux) ->
decode; parity of every frame hash against the ES upload path.  Kernel times: run under rocprofv3 --kernel-trace --stats.
    python tools/ts_bench.py [streams] [frames]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from jsmpeg_amd import batch as jb, synth  # noqa: E402

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 120
cfg = synth.CONFIGS[bench.CONFIG]
gen = bench.generate_streams(0, n_streams, frames)
es = [g[0] for g in gen]
ts = [synth.mux_ts(g[0], g[1]) for g in gen]
total_es, total_ts = sum(len(s) for s in es), sum(len(s) for s in ts)
with jb.Batch(cfg["width"], cfg["height"], n_streams, n_streams * frames + 8, total_es + 64 * n_streams + 4096) as b:
    b.upload(es)
    n = b.decode()
    want = b.frame_hashes().copy()
    t0 = time.perf_counter()
    b.upload_ts(ts)
    t_up = time.perf_counter() - t0
    assert b.decode() == n
    got = b.frame_hashes()
    assert np.array_equal(got, want), "TS path and ES path decode differently"
    assert all(len(b.ts_writes(s)) == frames for s in range(n_streams))
    print({"streams": n_streams, "ts_MB": round(total_ts / 1e6, 1), "es_MB": round(total_es / 1e6, 1),
           "upload_ts_ms_incl_pcie": round(t_up * 1e3, 1), "frames_equal": True})
