"""Renderer stage throughput: k_rgba over the frames a batch decode left in HBM (1080p: 3.13 MB read + 8.29 MB
written per frame).  python tools/rgba_bench.py [streams] [frames]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from jsmpeg_amd import batch as jb, synth  # noqa: E402

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 120
cfg = synth.CONFIGS[bench.CONFIG]
w, h = cfg["width"], cfg["height"]
streams = [g[0] for g in bench.generate_streams(0, n_streams, frames)]
total = sum(len(s) for s in streams)
with jb.Batch(w, h, n_streams, n_streams * frames + 8, total + 64 * n_streams + 4096) as b:
    b.upload(streams)
    n = b.decode()
    out = torch.empty((n, h, w, 4), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    b.render_rgba_device(0, n, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        b.render_rgba_device(0, n, out.data_ptr(), st.cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = n * (b.luma_bytes + 2 * b.chroma_bytes + w * h * 4)
    print({"frames": n, "ms": round(ms, 3), "frames_per_s": round(n / ms * 1e3), "GB_per_s": round(nbytes / ms / 1e6, 1),
           "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000, 3)})
