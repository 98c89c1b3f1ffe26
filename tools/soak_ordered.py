"""SOAK of the ordered reconstruct launch (one launch per batch, a picture's tiles wait for its forward reference through a
relaxed counter: kernels.hip jm_recon_wait -- the product default where a batch fills eight classes).  The design argues it
needs no fences (producer and consumer share an L2, no CU holds a line of a frame before the frame is complete); the tests
found the one case that argument missed (frame padding).  This is the record the argument did not have: many passes of
cfg2-SHAPED batches (64 streams: classes walk streams) and NARROW ones (one / four streams: classes walk GOP chains), fresh
content every few passes, each pass's hash vector against the same streams reconstructed LEVEL BY LEVEL (a batch created
with JSMPEG_HIP_RECON_ORDER=0: one launch per dependency level, stream-ordered -- no waits inside a launch), while
  * a second PROCESS streams HBM on the same device (its own 64 x 24 batch decoded and hashed -- k_hash over 4.8 GB -- back to back), and
  * a second BATCH is in flight on the same device from a second host thread (its own HIP stream, ordered too).
Reports passes, pictures compared, mismatches, the ordered launches' waits and status words.

    python tools/soak_ordered.py [--passes 2000] [--out profiles/r06_soak.txt] [--no-load]
"""
import argparse
import ctypes
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = (
    # (name, config, streams, pictures per stream, generator overrides, passes per content)
    ("cfg2-shaped 64 x 24 1080p (classes walk streams)", "cfg2_1080p", 64, 24, {}, 25),
    ("cfg2-shaped 16 x 36 1080p", "cfg2_1080p", 16, 36, {}, 25),
    ("coherent motion 64 x 24 1080p (unwritten macroblocks: stale waits)", "cfg2_1080p", 64, 24, dict(mv_jitter=1, f_code_max=1, coded_permille=60, ac_max=1), 25),
    ("narrow: one 720p stream x 240 (classes walk GOP chains)", "cfg1_720p", 1, 240, {}, 25),
    ("narrow: four 1080p streams x 96", "cfg2_1080p", 4, 96, {}, 25),
)


def hbm_load(seconds, device):
    """the second process: a 64 x 24 1080p batch decoded and hashed back to back (parse, reconstruct, k_hash over its 4.8 GB of
    frames) -- HBM traffic of the same kinds beside the soak's kernels, from another process"""
    import torch  # noqa: F401  (binds the HIP runtime torch ships)
    from jsmpeg_amd import batch as jb, synth
    streams = [synth.generate_config("cfg2_1080p", n_frames=24, stream=500 + s)[0] for s in range(64)]
    with jb.Batch(1920, 1080, 64, 64 * 24 + 8, sum(len(s) for s in streams) + 8192, device=device) as b:
        b.upload(streams)
        n, t0 = 0, time.time()
        while time.time() - t0 < seconds:
            b.decode()
            b.frame_hashes()
            n += 1
    print("load process: %d decode + hash rounds in %.0f s" % (n, seconds), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=2000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-load", action="store_true")
    ap.add_argument("--load-child", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    if a.load_child:
        return hbm_load(a.load_child, a.device)

    import torch
    from jsmpeg_amd import batch as jb, synth
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    say("soak of the ordered reconstruct: %d passes over %d shapes; product library %s" % (a.passes, len(SHAPES), os.path.basename(jb._build.LIB_HIP)))
    per_shape = max(1, a.passes // len(SHAPES))
    load = None
    if not a.no_load:
        load = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--load-child", "100000", "--device", str(a.device)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        say("second process streaming the same device: pid %d (decode + k_hash rounds, back to back)" % load.pid)
    total_pics = total_bad = total_passes = total_waits = 0
    flagged, redone = [], [0]
    t_all = time.time()
    try:
        for name, config, n_streams, frames, over, per_content in SHAPES:
            cfg = synth.CONFIGS[config]
            W, H = cfg["width"], cfg["height"]
            os.environ["JSMPEG_HIP_RECON_ORDER"] = "0"
            ref = None
            seed, done, bad, waits, launches_seen, t0 = 0, 0, 0, 0, set(), time.time()
            streams = None
            stop = threading.Event()
            other_err = []

            def second_batch():
                # a second batch in flight on the same device: its own stream, ordered launches of its own
                try:
                    s2 = torch.cuda.Stream(device=a.device)
                    os.environ.pop("JSMPEG_HIP_RECON_ORDER", None)
                    with jb.Batch(W, H, n_streams, n_streams * frames + 8, sum(len(s) for s in streams) + 64 * n_streams + 4096, device=a.device) as b2:
                        b2.upload(streams)
                        while not stop.is_set():
                            b2.decode(stream=ctypes.c_void_p(s2.cuda_stream))
                except Exception as e:   # noqa: BLE001
                    other_err.append(repr(e))

            while done < per_shape:
                streams = [synth.generate_config(config, n_frames=frames, stream=10000 * (SHAPES.index((name, config, n_streams, frames, over, per_content)) + 1) + 97 * seed + s, **over)[0]
                           for s in range(n_streams)]
                seed += 1
                cap_es = sum(len(s) for s in streams) + 64 * n_streams + 4096
                os.environ["JSMPEG_HIP_RECON_ORDER"] = "0"
                with jb.Batch(W, H, n_streams, n_streams * frames + 8, cap_es, device=a.device) as b0:     # level by level: the reference
                    b0.upload(streams)
                    b0.decode()
                    want = b0.frame_hashes().copy()
                    assert b0.recon_info()["group"] == 0
                os.environ.pop("JSMPEG_HIP_RECON_ORDER", None)
                stop.clear()
                th = threading.Thread(target=second_batch)
                th.start()
                with jb.Batch(W, H, n_streams, n_streams * frames + 8, cap_es, device=a.device) as b1:     # the product default
                    b1.upload(streams)
                    for _ in range(min(per_content, per_shape - done)):
                        b1.decode()
                        got = b1.frame_hashes()
                        info = b1.recon_info()
                        launches_seen.add((info["launches"], info["group"]))
                        waits += info["waits"]
                        if info["status"] & 3:                 # a wait ran out of patience / a class ran on two XCDs: the launch flagged itself
                            flagged.append((name, done, info["status"]))
                        elif info["status"] == 4:              # narrow batch whose GOP-chain assumption did not hold for this content: done over, by design
                            redone[0] += 1
                        n_bad = int(np.count_nonzero(got != want))
                        bad += n_bad
                        total_pics += len(want)
                        done += 1
                stop.set()
                th.join()
                if other_err:
                    raise RuntimeError("the second batch failed: " + other_err[0])
            total_bad += bad
            total_passes += done
            total_waits += waits
            say("%-72s %5d passes, %3d contents, %8d pictures compared, %d mismatches; launches per pass / lockstep %s; waits that found their picture unfinished: %d; %.0f s"
                % (name, done, seed, done * n_streams * frames, bad, sorted(launches_seen), waits, time.time() - t0))
    finally:
        if load:
            load.kill()
            load.wait()
    say("TOTAL: %d passes, %d pictures compared against the level-by-level reconstruct, %d MISMATCHES, %d launches flagged themselves %s, %d narrow passes done over by design (status 4), "
        "%d first looks found their picture unfinished; %.0f s" % (total_passes, total_pics, total_bad, len(flagged), flagged[:5] if flagged else "", redone[0], total_waits, time.time() - t_all))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write("\n".join(lines) + "\n")
    return 1 if total_bad or flagged else 0


if __name__ == "__main__":
    sys.exit(main() or 0)
