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


def measure_live(n_streams=64, n_frames=60, abi_streams=4, device=-1):
    """LIVE audio streams (include/jsmpeg_hip.h part 6): the same streams arriving a frame per stream per tick -- a write() per
    stream, ONE jsmpeg_hip_mp2_live_tick for all of them, the samples brought to the host -- beside the reference's one-frame
    decoder ABI driven the same way (a decoder per stream: write a frame, decode(), the samples on the host).  Every frame of
    every tick against the oracle's decode of the whole stream."""
    from jsmpeg_amd import build, cabi, mp2, synth
    made = [synth.generate_mp2_config("mp2_stereo_44k_192", n_frames, stream=s) for s in range(n_streams)]
    oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    want = [cabi.decode_mp2_stream(oracle, d)[0] for d, _ in made]
    bounds = [[int(o) for o in offs] + [len(d)] for d, offs in made]
    t_tick, t_write, t_read, parts, bad = [], [], [], [], 0
    with mp2.Mp2Live(n_streams, max_frames_per_tick=2, device=device) as live:
        for _ in range(n_streams):
            live.open()
        for k in range(n_frames):
            pieces = [made[s][0][bounds[s][k]:bounds[s][k + 1]] for s in range(n_streams)]
            t0 = time.perf_counter()
            for s in range(n_streams):
                live.write(s, k * 1152 / 44100.0, pieces[s])
            t1 = time.perf_counter()
            n = live.tick()
            t2 = time.perf_counter()
            pcm = live.read_pcm()
            t3 = time.perf_counter()
            if n != n_streams:
                raise RuntimeError("live audio tick %d decoded %d frames, expected %d" % (k, n, n_streams))
            for s in range(n_streams):                 # the checker, outside the clocks
                bad += not np.array_equal(pcm[s].view(np.uint32), want[s][k].view(np.uint32))
            if k >= 3:
                t_write.append(t1 - t0); t_tick.append(t2 - t1); t_read.append(t3 - t2); parts.append(live.timings())
    if bad:
        raise RuntimeError("live audio parity failure: %d frames differ from the oracle" % bad)
    # the one-frame ABI the same way: abi_streams decoders in turn
    t_abi = []
    decs = [cabi.Mp2Decoder(build.LIB_HIP, 128 * 1024, cabi.MODE_EVICT) for _ in range(abi_streams)]
    try:
        for k in range(n_frames):
            for s in range(abi_streams):
                piece = made[s][0][bounds[s][k]:bounds[s][k + 1]]
                t0 = time.perf_counter()
                decs[s].write(piece)
                if not decs[s].decode():
                    raise RuntimeError("the one-frame ABI decoded nothing")
                decs[s].channels()
                if k >= 3:
                    t_abi.append(time.perf_counter() - t0)
    finally:
        for d in decs:
            d.close()
    med = lambda v: float(np.median(v)) * 1e3
    tick_ms, write_ms, read_ms, abi_ms = med(t_tick), med(t_write), med(t_read), med(t_abi)
    whole = write_ms + tick_ms + read_ms
    return {"streams": n_streams, "frames_per_stream_per_tick": 1, "ticks": n_frames, "frames_differing_from_oracle": 0,
            "ms_per_tick": round(whole, 4), "ms_writes": round(write_ms, 4), "ms_tick_call": round(tick_ms, 4), "ms_samples_to_host": round(read_ms, 4),
            "parts_ms": {k: round(float(np.median([p[k] for p in parts])), 4) for k in parts[0]},
            "frames_per_s": round(n_streams / (whole * 1e-3), 1), "realtime_streams": round(n_streams / (whole * 1e-3) * 1152 / 44100, 1),
            "one_frame_abi": {"ms_per_frame": round(abi_ms, 4), "frames_per_s": round(1e3 / abi_ms, 1),
                              "note": "mp2_decoder_* of the same library, %d decoders in turn: write a frame, decode(), the samples on the host" % abi_streams},
            "live_over_one_frame_abi": round((n_streams / whole) / (1.0 / abi_ms), 2),
            "note": "tools/mp2_bench.py measure_live: every tick = one write() per stream (a whole frame) + ONE jsmpeg_hip_mp2_live_tick + the "
                    "tick's samples to the host in one call, host clock, median; every frame of every tick == the oracle's"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--frames", type=int, default=154)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--live", action="store_true", help="the live streams' tick instead (measure_live)")
    a = ap.parse_args()
    print(json.dumps(measure_live(a.streams, min(a.frames, 60)) if a.live else measure(a.streams, a.frames, a.reps)))
