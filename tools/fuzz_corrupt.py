"""Robustness sweep (results unspecified, like the reference's on damaged input -- but no fault, no hang): valid synthetic
streams with random byte damage, truncation and spliced garbage through the batch interface and the one-picture ABI.
    python tools/fuzz_corrupt.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import batch as jb, build, cabi, synth  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
for c in range(cases):
    w, h = int(rng.integers(2, 20)) * 16, int(rng.integers(2, 14)) * 16
    n = int(rng.integers(3, 12))
    es, offs = synth.generate_config("cfg1_720p", n_frames=n, stream=c, width=w, height=h, gop=int(rng.choice([1, 4, 12])),
                                  ac_max=int(rng.choice([1, 8, 40])), syntax_quirks=int(rng.integers(0, 2)), f_code_max=int(rng.integers(1, 8)))
    bad = es.copy()
    kind = int(rng.integers(0, 4))
    if kind == 0:      # random byte damage, 0.1 % .. 5 % of the bytes
        k = max(1, int(len(bad) * float(rng.choice([0.001, 0.01, 0.05]))))
        bad[rng.integers(0, len(bad), k)] = rng.integers(0, 256, k, dtype=np.uint8)
    elif kind == 1:    # truncation at a random point
        bad = bad[:int(rng.integers(8, len(bad)))].copy()
    elif kind == 2:    # runs of zeros / ones / garbage spliced in
        for _ in range(int(rng.integers(1, 6))):
            at, ln = int(rng.integers(0, len(bad))), int(rng.integers(1, 400))
            bad[at:at + ln] = int(rng.choice([0, 255])) if rng.integers(0, 2) else rng.integers(0, 256, len(bad[at:at + ln]), dtype=np.uint8)
    else:              # start codes sprinkled at random places
        for _ in range(int(rng.integers(1, 30))):
            at = int(rng.integers(0, len(bad) - 4))
            bad[at:at + 4] = [0, 0, 1, int(rng.integers(0, 256))]
    with jb.Batch(w, h, 2, 2 * n + 64, len(bad) + len(es) + 8192) as b:
        b.upload([bad, es])
        got = b.decode()
        b.frame_hashes()
    # the same for the ingest stage: a packet-aligned TS with damaged packet contents (sync bytes kept or not)
    ts = synth.mux_ts(es, offs)
    k_ts = max(1, len(ts) // 50)
    ts[rng.integers(0, len(ts), k_ts)] = rng.integers(0, 256, k_ts, dtype=np.uint8)
    if rng.integers(0, 2):
        ts[::188] = 0x47
    with jb.Batch(w, h, 1, 2 * n + 64, len(ts) + 8192) as b:
        try:
            b.upload_ts([ts])
            b.decode()
        except RuntimeError:
            pass      # a packet without sync byte / a header longer than its packet / too many PIDs: reported, not decoded
    with cabi.Mpeg1Decoder(build.LIB_HIP, len(bad) + 1024, cabi.MODE_EXPAND) as d:
        d.write(bad)
        k = 0
        while k < 4 * n + 8 and d.decode():
            k += 1
    if os.environ.get("FUZZ_VERBOSE"):
        print("case %d kind %d %dx%d n=%d: batch found %d pictures, one-picture ABI decoded %d" % (c, kind, w, h, n, got, k), flush=True)
print("%d damaged streams decoded without fault or hang in %.1fs" % (cases, time.time() - t0))
