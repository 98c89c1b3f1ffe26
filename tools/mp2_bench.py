#!/usr/bin/env python3
"""Throughput of the MP2 audio stage on the audio that goes with the video benchmark batch:
64 stereo 44.1 kHz 192 kbit/s Layer II streams x 154 frames (the 4 s that 120 pictures at 30 fps last), resident
in HBM, decoded by jsmpeg_hip_mp2_batch_decode; timed by the engine's HIP events on the launch stream.  Parity:
a sample of streams bit for bit against the oracle.  Beside it the reference's own C decoder (oracle/_ref) on one
host core.  bench.py calls measure() after its timed region and attaches the result as "audio_stage"; standalone:

    python tools/mp2_bench.py [--streams 64] [--frames 154] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def measure(n_streams=64, n_frames=154, reps=5, cpu_baseline=True, device=-1):
    from jsmpeg_amd import build, cabi, mp2, synth
    streams = [synth.generate_mp2_config("mp2_stereo_44k_192", n_frames, stream=s)[0] for s in range(n_streams)]
    es_bytes = int(sum(len(s) for s in streams))
    oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    with mp2.Mp2Batch(n_streams, es_bytes + 64, device) as b:
        b.upload(streams)
        total = b.decode()                      # warm-up (allocations, code load)
        if total != n_streams * n_frames:
            raise RuntimeError("MP2 batch decoded %d frames, expected %d" % (total, n_streams * n_frames))
        t = []
        for _ in range(reps):
            b.decode()
            t.append(b.timings())
        checked = sorted(set([0, n_streams // 2, n_streams - 1]))
        for s in checked:                       # the checker, after the timed decodes
            want = cabi.decode_mp2_stream(oracle, streams[s])[0]
            got = b.read_pcm(s)
            if got.shape != want.shape or not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                raise RuntimeError("MP2 parity failure on stream %d" % s)
    med = {k: float(np.median([x[k] for x in t])) for k in t[0]}
    pcm_bytes = total * 2 * 1152 * 4
    alg = es_bytes + pcm_bytes                  # compressed bytes read once + PCM written once
    out = {
        "metric": "MP2 (MPEG-1 Audio Layer II) decode throughput", "value": round(total / (med["total_ms"] * 1e-3), 1),
        "unit": "frames/s", "dtype": "f32 (+ f64 products, int32 accumulator: the reference C's arithmetic)",
        "workload": "%d streams x %d frames, stereo 44.1 kHz 192 kbit/s, batched on one GPU" % (n_streams, n_frames),
        "ms_per_pass": round(med["total_ms"], 4), "phases_ms": {k: round(v, 4) for k, v in med.items()},
        "realtime_streams": round(total / (med["total_ms"] * 1e-3) * 1152 / 44100, 1),
        "roofline": {"bound": "hbm", "achieved": round(alg / (med["total_ms"] * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(alg / (med["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "algorithmic_bytes_per_pass": alg,
                     "note": "whole pass (frame walk, host turn-around, side information, matrixing, windowing); "
                             "%.1f MB per pass: launch- and latency-bound at this size, not HBM-bound" % (alg / 1e6)},
        "parity_checked": "streams %s bit for bit against the oracle" % checked,
    }
    if cpu_baseline:
        lib = build.LIB_REF if os.path.exists(build.LIB_REF) else oracle
        runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            n = 0
            for s in streams:
                n += len(cabi.decode_mp2_stream(lib, s)[0])
            runs.append(time.perf_counter() - t0)
        dt = sorted(runs)[1]
        out["cpu_baseline"] = {"value": round(n / dt, 1), "unit": "frames/s", "cores": 1,
                               "kind": "reference" if lib == build.LIB_REF else "port",
                               "sample": "all the streams (%d frames), decoded one after another on one core, median of 3, %s"
                                         % (n, "reference src/wasm/mp2.c gcc -O3" if lib == build.LIB_REF else "oracle/mp2_oracle.c")}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--frames", type=int, default=154)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    print(json.dumps(measure(a.streams, a.frames, a.reps)))
