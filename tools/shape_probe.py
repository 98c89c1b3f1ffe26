"""Batches of other shapes than the benchmark's, as timings + parity of a few streams against the oracle (checker):
all-intra, long GOPs, ragged stream lengths.   python tools/shape_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import batch as jb, build, cabi, hashing, synth  # noqa: E402

LIB_ORACLE = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()


def oracle_hashes(es):
    out = []
    with cabi.Mpeg1Decoder(LIB_ORACLE, len(es) + 1024, cabi.MODE_EXPAND) as dec:
        dec.write(es)
        while dec.decode():
            out.append(hashing.frame_hash(*dec.planes()))
    return out


def run(name, specs, reps=3):
    """specs: list of (n_frames, overrides) per stream, 1080p"""
    streams = [synth.generate_config("cfg2_1080p", n_frames=n, stream=1000 + k, **ov)[0] for k, (n, ov) in enumerate(specs)]
    total = sum(len(s) for s in streams)
    n_pics = sum(n for n, _ in specs)
    with jb.Batch(1920, 1080, len(streams), n_pics + 8, total + 64 * len(streams) + 4096) as b:
        b.upload(streams)
        acc = None
        for r in range(reps):
            b.decode()
            t = b.timings()
            if r:
                acc = t if acc is None else {k: acc[k] + t[k] for k in t}
        c = b.counters()
        ms = acc["total_ms"] / (reps - 1)
        # parity: first and last stream, every picture, device hash against the oracle's planes
        hashes = b.frame_hashes()
        first = 0
        ok = True
        for k, s in enumerate(streams):
            n = specs[k][0]
            if k in (0, len(streams) - 1):
                want = oracle_hashes(s)
                ok = ok and [int(x) for x in hashes[first:first + n]] == [int(x) for x in want]
            first += n
        print("%-34s %4d streams %6d pictures %3d levels: %7.2f ms  (index %.2f host %.2f parse %.2f recon %.2f)  %8.0f frames/s  parity %s"
              % (name, len(streams), n_pics, c["levels"], ms, acc["index_ms"] / (reps - 1), acc["host_ms"] / (reps - 1), acc["parse_ms"] / (reps - 1), acc["recon_ms"] / (reps - 1), n_pics / ms * 1e3,
                 "ok" if ok else "MISMATCH"))


run("all intra (GOP 1)", [(60, dict(gop=1))] * 64)
run("all intra (GOP 1), 5 reps", [(60, dict(gop=1))] * 64, reps=5)
run("GOP 60", [(120, dict(gop=60))] * 64)
run("GOP 12, coherent motion (pan +- 1)", [(120, dict(mv_jitter=2))] * 64)
run("ragged: 1 x 1200 + 63 x 12", [(1200, {})] + [(12, {})] * 63)
run("8 streams x 120", [(120, {})] * 8)
