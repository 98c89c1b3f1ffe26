"""On the GPU box: the batch path on 1080p content made by the test-side ENCODER (tests/enc/mpeg1_enc.py through
tools/enc_content.py: GOPs of 12 pictures with coded-video statistics -- coherent vector fields, zero vectors, skipped runs,
sparse residuals, intra pictures 10-30 x the predicted ones), beside the headline's uniform-random syntax.  64 streams x 120
pictures: stream s = the GOPs in the rotation that starts at GOP s % n (every GOP begins with an intra picture and its own
sequence header, of which the decoder reads the first: mpeg1.c:814), so there are n distinct streams, and every one of them is
held against the oracle picture by picture before a figure is printed.  The same streams lie at 64 different places of the
batch: nothing is shared between them on the device.
    python tools/enc_content_bench.py [streams] [GOPs per stream] [reps] [--out file.json] [--gops all | 0,2,4,6] [--two] [--node]
--gops: default = the four committed under tests/golden/enc1080/ (GOPs 0, 2, 4, 6 of tools/enc_content.py: quantiser 6-10, the
        short search range: ~16 Mbit/s per stream, the headline's bit rate; golden vectors beside them); `all`: also those of
        tests/enc/_cache/ (all eight: ~39 Mbit/s, intra pictures of 0.43-0.77 MB)
--two:  also two batch objects decoded side by side (bench.py two_batches_in_flight: one's slice parse beside the other's
        reconstruct -- on this content the parse is as long as its longest slices' walk, and the other batch fills the GPU meanwhile)
--node: also the same streams from Node (tools/bench_node.js --two: JSMpeg.HIPBatch, one batch at a time and two in flight by decodeAsync())
bench.py attaches run() as `coded_video_content`."""
import ctypes
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H = 1920, 1080
END = bytes([0, 0, 1, 0xB7])


def gop_files(which=None):
    """{GOP number: path}: the committed four, and (which == "all" or a list naming others) the cache's"""
    out = {}
    for d in (os.path.join(ROOT, "tests", "enc", "_cache"), os.path.join(ROOT, "tests", "golden", "enc1080")):   # (the committed copy wins)
        for f in glob.glob(os.path.join(d, "enc1080_*.m1v")):
            out[int(os.path.basename(f)[len("enc1080_"):-len(".m1v")])] = f
    committed = sorted(int(os.path.basename(f)[len("enc1080_"):-len(".m1v")]) for f in glob.glob(os.path.join(ROOT, "tests", "golden", "enc1080", "enc1080_*.m1v")))
    keep = committed if which is None else (sorted(out) if which == "all" else list(which))
    missing = [k for k in keep if k not in out]
    if missing or not keep:
        raise RuntimeError("encoder GOPs %s not found (tests/golden/enc1080/, tests/enc/_cache/: tools/enc_content.py makes them)" % (missing or "(none)"))
    return {k: out[k] for k in keep}


def run(n_streams=64, gops_per_stream=10, reps=8, which=None, two=False, device=-1, two_fn=None, napi_fn=None):
    from jsmpeg_amd import batch as jb, build, cabi, hashing
    files = gop_files(which)
    gops, n_pics_gop, intra_bytes = [], [], []
    for k in sorted(files):
        es = np.fromfile(files[k], dtype=np.uint8)
        assert bytes(es[-4:]) == END, files[k]
        g = es[:-4]                       # (the sequence end code goes behind the stream's last GOP)
        at = np.flatnonzero((g[:-3] == 0) & (g[1:-2] == 0) & (g[2:-1] == 1) & (g[3:] == 0))
        gops.append(g)
        n_pics_gop.append(int(at.size))
        intra_bytes.append(int(at[1] - at[0]) if at.size > 1 else int(len(g)))

    def stream_of(first):
        return np.concatenate([gops[(first + k) % len(gops)] for k in range(gops_per_stream)] + [np.frombuffer(END, np.uint8)])
    distinct = [stream_of(k) for k in range(min(len(gops), n_streams))]
    streams = [distinct[s % len(distinct)] for s in range(n_streams)]
    pics_per_stream = [sum(n_pics_gop[(s + k) % len(gops)] for k in range(gops_per_stream)) for s in range(len(distinct))]
    total = sum(len(s) for s in streams)
    n_pictures = sum(pics_per_stream[s % len(distinct)] for s in range(n_streams))

    # the checker: the oracle, picture by picture (tests / bench gates only: the timed path never touches it); it also counts
    # SURVEY.md 8d's predicted macroblocks (its copy_macroblock calls: skipped macroblocks of P pictures included)
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
    if [len(h) for h in want] != pics_per_stream:
        raise RuntimeError("the oracle decoded %r pictures per stream, the streams hold %r" % ([len(h) for h in want], pics_per_stream))
    want_all = [want[s % len(distinct)] for s in range(n_streams)]

    def make_batch():
        return jb.Batch(W, H, n_streams, n_pictures + 8, total + 64 * n_streams + 4096, device=device)

    two_res = None
    with make_batch() as b:
        b.upload(streams)
        acc, warm = None, 2
        for r in range(reps + warm):
            n = b.decode()
            if n != n_pictures:
                raise RuntimeError("decoded %d pictures, expected %d" % (n, n_pictures))
            t = b.timings()
            if r >= warm:
                acc = t if acc is None else {k: acc[k] + t[k] for k in t}
        dev = b.frame_hashes()
        per = {}
        for p, i in enumerate(b.pictures()):
            per.setdefault(i.stream, []).append(int(dev[p]))
        bad = [s for s in range(n_streams) if per.get(s, []) != want_all[s]]
        if bad:
            raise RuntimeError("PARITY FAILURE against the oracle on streams %s" % bad[:8])
        info = b.recon_info()
        counters = b.counters()
        if two:
            if two_fn is None:
                import bench  # noqa: E402  (two_batches_in_flight only)
                two_fn = bench.two_batches_in_flight
            two_res = two_fn(b, make_batch, lambda bb, sptr: bb.upload(streams), n_pictures, want_all, passes=8)
            two_res.pop("note", None)
    ms = {k: round(v / reps, 4) for k, v in acc.items()}
    mb = ((W + 15) // 16) * ((H + 15) // 16)
    predicted = sum(predicted_of[s % len(distinct)] for s in range(n_streams))
    alg = 384 * mb * n_pictures + 384 * predicted          # 384 bytes written per macroblock + 384 read per predicted one
    res = {
        "workload": "%d streams x %d pictures 1920x1080, encoder-made content (GOPs %s of tools/enc_content.py, %s pictures each, rotated per stream: %d distinct streams)"
                    % (n_streams, pics_per_stream[0], sorted(files), sorted(set(n_pics_gop)), len(distinct)),
        "content": "procedural moving pictures through the test-side block-matching encoder (tests/enc/mpeg1_enc.py: full-pel search + half-pel "
                   "refinement, DCT, quantiser, skipped and not-coded macroblocks, intra fallback): coherent vector fields, sparse residuals, intra "
                   "pictures many times the predicted ones",
        "es_bytes": total, "mbit_per_s_per_stream_at_30fps": round(total / n_streams * 8 / (pics_per_stream[0] / 30.0) / 1e6, 2),
        "bytes_per_macroblock": round(total / n_pictures / mb, 2), "gop_bytes": [int(len(g)) for g in gops], "intra_picture_bytes": intra_bytes,
        "pictures": n_pictures, "ms_per_pass": ms["total_ms"], "frames_per_s": round(n_pictures / ms["total_ms"] * 1e3, 1),
        "mpixel_per_s": round(n_pictures / ms["total_ms"] * 1e3 * W * H / 1e6, 1), "gpu_phases_ms": ms,
        "macroblocks": mb * n_pictures, "predicted_macroblocks": int(predicted),
        "k_recon_algorithmic_bytes": int(alg), "k_recon_frac_of_8TBs": round(alg / (ms["recon_ms"] * 1e-3) / 8e12, 4),
        "whole_step_frac_of_8TBs": round((alg + total) / (ms["total_ms"] * 1e-3) / 8e12, 4),
        "reconstruct": info, "levels": counters.get("levels"), "uncovered_pictures": counters.get("uncovered_pictures"),
        "parity": "all %d streams, every picture: device hash == oracle (%d distinct streams decoded by the oracle in %.1f s; the committed GOPs' "
                  "golden vectors: tests/golden/enc1080/, reference JS == wasm == C == oracle)" % (n_streams, len(distinct), oracle_s),
        "clock": "the engine's phase events, mean of %d passes after %d warm-up passes" % (reps, warm),
        "note": "the slice parse is one lane per slice and lasts as long as the longest slices' walk: an intra picture of coded video is 10-30 x a "
                "predicted one, so the pass is the intra slices' serial walk (~1 us per byte) while most SIMDs stand idle -- which a second batch "
                "in flight fills (two_batches_in_flight); the headline's generator content has intra pictures ~2 x the predicted ones",
    }
    if two_res is not None:
        res["two_batches_in_flight"] = two_res
    if napi_fn is not None:       # the same streams from Node (bench.py via_napi: JSMpeg.HIPBatch over the N-API addon, one batch and two in flight)
        try:
            r = napi_fn(streams, {s: want_all[s] for s in range(n_streams)}, W, H, pics_per_stream[0], max(2, reps), 2, max(0, device))
            for k in ("value", "ms_per_step"):
                if isinstance(r.get(k), float):
                    r[k] = round(r[k], 3)
            t = r.get("two_batches_in_flight") or {}
            for k in ("value", "ms_per_pass", "passes_in_window"):
                if isinstance(t.get(k), float):
                    t[k] = round(t[k], 3)
            res["via_napi"] = r
        except Exception as e:   # noqa: BLE001 (an extra of an extra)
            res["via_napi"] = {"error": repr(e)[:300]}
    return res


if __name__ == "__main__":
    argv = sys.argv[1:]
    out_path = argv[argv.index("--out") + 1] if "--out" in argv else None
    sel = argv[argv.index("--gops") + 1] if "--gops" in argv else None
    which = None if sel is None else ("all" if sel == "all" else [int(x) for x in sel.split(",")])
    two = "--two" in argv
    if two:
        argv.remove("--two")
    node = "--node" in argv
    if node:
        argv.remove("--node")
    for opt in ("--out", "--gops"):
        if opt in argv:
            i = argv.index(opt)
            del argv[i:i + 2]
    napi_fn = None
    if node:
        import bench  # noqa: E402  (via_napi only)
        napi_fn = bench.via_napi
    res = run(int(argv[0]) if len(argv) > 0 else 64, int(argv[1]) if len(argv) > 1 else 10, int(argv[2]) if len(argv) > 2 else 8, which, two, napi_fn=napi_fn)
    print(json.dumps(res))
    if out_path:
        with open(out_path, "w") as f:
            json.dump(res, f, indent=1)
