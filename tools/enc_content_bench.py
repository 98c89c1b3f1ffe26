"""On the GPU box: the batch path on 1080p content made by the test-side ENCODER (tools/enc_content.py -> tests/enc/_cache/:
eight GOPs of 12 pictures with coded-video statistics -- coherent vector fields, zero vectors, skipped runs, sparse
high frequencies), beside the headline's uniform-random syntax.  64 streams x 120 pictures: stream s = the GOPs in the
rotation that starts at GOP s % 8 (every GOP begins with an intra picture and its own sequence header, of which the
decoder reads the first: mpeg1.c:814), so there are eight distinct streams and every one of them is held against the
oracle picture by picture before a figure is printed.  The same streams lie at 64 different places of the batch: nothing
is shared between them on the device.
    python tools/enc_content_bench.py [streams] [GOPs per stream] [reps] [--out file.json]"""
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import batch as jb, build, cabi, hashing  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
n_streams = int(args[0]) if len(args) > 0 else 64
gops_per_stream = int(args[1]) if len(args) > 1 else 10
reps = int(args[2]) if len(args) > 2 else 8
W, H = 1920, 1080

files = sorted(glob.glob(os.path.join(ROOT, "tests", "enc", "_cache", "enc1080_*.m1v")))
if not files:
    raise SystemExit("no tests/enc/_cache/enc1080_*.m1v: run tools/enc_content.py first (CPU, minutes)")
END = bytes([0, 0, 1, 0xB7])
gops = []
for f in files:
    es = np.fromfile(f, dtype=np.uint8)
    assert bytes(es[-4:]) == END, f
    gops.append(es[:-4])          # (the sequence end code goes behind the stream's last GOP)
n_pics_gop = []
for g in gops:
    n_pics_gop.append(int(np.count_nonzero((g[:-3] == 0) & (g[1:-2] == 0) & (g[2:-1] == 1) & (g[3:] == 0))))


def stream_of(first):
    parts = [gops[(first + k) % len(gops)] for k in range(gops_per_stream)]
    return np.concatenate(parts + [np.frombuffer(END, np.uint8)])


distinct = [stream_of(k) for k in range(min(len(gops), n_streams))]
streams = [distinct[s % len(distinct)] for s in range(n_streams)]
pics_per_stream = [sum(n_pics_gop[(s + k) % len(gops)] for k in range(gops_per_stream)) for s in range(len(distinct))]
total = sum(len(s) for s in streams)
n_pictures = sum(pics_per_stream[s % len(distinct)] for s in range(n_streams))

# the checker: the oracle, picture by picture (tests / bench gates only: the timed path never touches it)
import ctypes  # noqa: E402
olib = cabi.load(build.LIB_ORACLE)
olib.oracle_debug_predicted_macroblocks.restype = ctypes.c_ulonglong
t0 = time.time()
want, predicted_of = [], []
for es in distinct:
    p0 = olib.oracle_debug_predicted_macroblocks()
    h = []
    with cabi.Mpeg1Decoder(build.LIB_ORACLE, len(es) + 1024, cabi.MODE_EXPAND) as dec:
        dec.write(es)
        while dec.decode():
            h.append(hashing.frame_hash(*dec.planes()))
    want.append(h)
    predicted_of.append(olib.oracle_debug_predicted_macroblocks() - p0)
oracle_s = time.time() - t0
assert [len(h) for h in want] == pics_per_stream, ([len(h) for h in want], pics_per_stream)

with jb.Batch(W, H, n_streams, n_pictures + 8, total + 64 * n_streams + 4096) as b:
    b.upload(streams)
    acc, warm = None, 2
    for r in range(reps + warm):
        n = b.decode()
        assert n == n_pictures, (n, n_pictures)
        t = b.timings()
        if r >= warm:
            acc = t if acc is None else {k: acc[k] + t[k] for k in t}
    dev = b.frame_hashes()
    per = {}
    for p, i in enumerate(b.pictures()):
        per.setdefault(i.stream, []).append(int(dev[p]))
    bad = [s for s in range(n_streams) if per.get(s, []) != want[s % len(distinct)]]
    if bad:
        raise SystemExit("PARITY FAILURE against the oracle on streams %s" % bad[:8])
    info = b.recon_info()
    counters = b.counters()
ms = {k: round(v / reps, 4) for k, v in acc.items()}
mb = ((W + 15) // 16) * ((H + 15) // 16)
# SURVEY.md 8d's bytes for the reconstruct: 384 written per macroblock + 384 read per PREDICTED macroblock (counted by the
# checker: the oracle's copy_macroblock calls on the same streams -- skipped macroblocks of P pictures included, like the generator's stats)
predicted = sum(predicted_of[s % len(distinct)] for s in range(n_streams))
alg = 384 * mb * n_pictures + 384 * predicted
res = {
    "workload": "%d streams x %d pictures 1920x1080, encoder-made content (tests/enc/mpeg1_enc.py: %d GOPs of %s pictures, rotated per stream)"
                % (n_streams, pics_per_stream[0], len(gops), sorted(set(n_pics_gop))),
    "content": "procedural moving pictures through a block-matching encoder (full-pel search + half-pel refinement, DCT, quantiser "
               "3-10, skipped and not-coded macroblocks): coherent vector fields and sparse residuals; %d distinct streams" % len(distinct),
    "es_bytes": total, "bytes_per_picture": round(total / n_pictures, 1), "mbit_per_s_per_stream_at_30fps": round(total / n_streams * 8 / (pics_per_stream[0] / 30.0) / 1e6, 2),
    "bytes_per_macroblock": round(total / n_pictures / mb, 2),
    "pictures": n_pictures, "ms_per_pass": ms["total_ms"], "frames_per_s": round(n_pictures / ms["total_ms"] * 1e3, 1),
    "mpixel_per_s": round(n_pictures / ms["total_ms"] * 1e3 * W * H / 1e6, 1), "gpu_phases_ms": ms,
    "macroblocks": mb * n_pictures, "predicted_macroblocks": predicted,
    "k_recon_algorithmic_bytes": alg, "k_recon_frac_of_8TBs": round(alg / (ms["recon_ms"] * 1e-3) / 8e12, 4),
    "whole_step_frac_of_8TBs": round((alg + total) / (ms["total_ms"] * 1e-3) / 8e12, 4),
    "reconstruct": info, "levels": counters.get("levels"), "uncovered_pictures": counters.get("uncovered_pictures"),
    "parity": "all %d streams, every picture: device hash == oracle (%d distinct streams decoded by the oracle in %.1f s)" % (n_streams, len(distinct), oracle_s),
    "clock": "the engine's phase events, mean of %d passes after %d warm-up passes" % (reps, warm),
}
print(json.dumps(res))
if out_path:
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
