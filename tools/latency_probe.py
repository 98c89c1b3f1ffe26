"""The other configurations of SURVEY.md 8d as timings (not bench lines): one 1280x720 stream of 360 pictures through the
batch interface (GOPs in parallel, 30 pictures per level), and the per-picture latency of the one-picture ABI (what the
Node class sits on) at 720p and 1080p, planes copied to the host each time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import batch as jb, build, cabi, synth  # noqa: E402

es, offs = synth.generate_config("cfg1_720p", n_frames=360)
with jb.Batch(1280, 720, 1, 368, len(es) + 4096) as b:
    b.upload([es])
    b.decode()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        b.decode()
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    print("cfg1 batch: one 1280x720 stream, 360 pictures: %.2f ms per pass on the host clock (median of 9; min %.2f max %.2f) = %.0f frames/s  %s"
          % (dt * 1e3, min(ts) * 1e3, max(ts) * 1e3, 360 / dt, b.timings()))
for name, cfg, n in (("720p", "cfg1_720p", 60), ("1080p", "cfg2_1080p", 120)):
    es, offs = synth.generate_config(cfg, n_frames=n)
    for ahead in (None, "16", "0"):
        os.environ.pop("JSMPEG_HIP_DECODE_AHEAD", None)
        if ahead is not None:
            os.environ["JSMPEG_HIP_DECODE_AHEAD"] = ahead        # read when a decoder is created (None: the default, by picture size)
        # (a) a BUFFERED stream: everything written, then decode() until false -- with decode-ahead the batch engine takes 16 pictures a pass
        with cabi.Mpeg1Decoder(build.LIB_HIP, len(es) + 1024, cabi.MODE_EXPAND) as d:
            d.write(es)
            t0 = time.perf_counter()
            d.decode()
            first = time.perf_counter() - t0
            t0 = time.perf_counter()
            k = 1
            while d.decode():
                k += 1
            dt = (time.perf_counter() - t0) / (k - 1)
            print("one-picture ABI %s, buffered stream, decode-ahead %s: %.3f ms per decode() incl. planes to the host (%d pictures) = %.0f frames/s; "
                  "first decode() %.3f ms; ahead (passes, served) %r" % (name, ahead, dt * 1e3, k, 1 / dt, first * 1e3, d.ahead_stats()))
    # (b) STREAMING: a picture written, a picture pulled (ts.js + Player): decode-ahead never engages
    os.environ.pop("JSMPEG_HIP_DECODE_AHEAD", None)
    with cabi.Mpeg1Decoder(build.LIB_HIP, 1 << 20, cabi.MODE_EVICT) as d:
        ts, k = [], 0
        for i in range(n):
            end = len(es) if i == n - 1 else int(offs[i + 1])
            d.write(es[int(offs[i]):end])
            t0 = time.perf_counter()
            while d.decode():
                k += 1
            ts.append(time.perf_counter() - t0)
        print("one-picture ABI %s, streaming (write a picture, decode): %.3f ms per picture (median of %d, %d decoded); ahead %r"
              % (name, sorted(ts)[len(ts) // 2] * 1e3, len(ts), k, d.ahead_stats()))
os.environ.pop("JSMPEG_HIP_DECODE_AHEAD", None)
# MP2 audio: the one-frame ABI (what JSMpeg.Decoder.MP2AudioHIP sits on), one decode() at a time with the PCM copied to the host
data, _ = synth.generate_mp2_config("mp2_stereo_44k_192", 400)
with cabi.Mp2Decoder(build.LIB_HIP, len(data) + 1024, cabi.MODE_EXPAND) as d:
    d.write(data)
    d.decode()
    t0 = time.perf_counter()
    k = 1
    while d.decode():
        k += 1
    dt = (time.perf_counter() - t0) / (k - 1)
    print("one-frame MP2 ABI: %.3f ms per decode() incl. PCM to the host (%d frames) = %.0f frames/s = %.0f x real time at 44.1 kHz"
          % (dt * 1e3, k, 1 / dt, 1152 / 44100 / dt))
