"""Robustness sweep (results unspecified, like the reference's on damaged input -- but no fault, no hang): valid synthetic
streams with random byte damage, truncation and spliced garbage through the batch interface, the one-picture ABI and the
live interface (there beside a GOOD stream in the same ticks, whose pictures must come out right regardless).
    python tools/fuzz_corrupt.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import batch as jb, build, cabi, hashing, live as jl, synth  # noqa: E402

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
    # live streams: the damaged ES in random pieces and the damaged TS through the library's demuxer, each beside the intact stream
    # written picture by picture -- ticks of both kinds, some in two halves; the intact stream's pictures are the whole stream's
    good_want = None
    with jb.Batch(w, h, 1, n + 8, len(es) + 8192) as b:
        b.upload([es])
        b.decode()
        hs = b.frame_hashes()
        good_want = [int(hs[p]) for p, info in enumerate(b.pictures()) if info.decoded]
    good_want = [x for k2, x in enumerate(good_want) if k2 == 0 or x != good_want[k2 - 1]]
    with jl.Live(w, h, 3, pictures_per_tick=16, store_bytes=2 * (len(es) + len(bad)) + 4096) as lv:
        ids = [lv.open() for _ in range(3)]
        good = [es[int(offs[k2]):(len(es) if k2 == n - 1 else int(offs[k2 + 1]))] for k2 in range(n)]
        at_bad, at_ts, got_good, k2 = 0, 0, [], 0
        for _ in range(20 * n + 50):                  # (bounded whatever the streams do)
            if k2 >= n and at_bad >= len(bad) and at_ts >= len(ts):
                break
            two = bool(rng.integers(0, 2))
            if two:
                lv.tick_begin(flush=bool(rng.integers(0, 2)))
            if k2 < n:
                lv.write(ids[0], good[k2])
                k2 += 1
            if at_bad < len(bad):
                ln = int(rng.integers(1, max(2, len(bad) // 3)))
                lv.write(ids[1], bad[at_bad:at_bad + ln])
                at_bad += ln
            if at_ts < len(ts):
                ln = int(rng.integers(1, max(2, len(ts) // 3)))
                try:
                    lv.write_ts(ids[2], ts[at_ts:at_ts + ln])
                except RuntimeError:
                    pass                              # a PES larger than the store (damaged lengths): refused, the demuxer moves on
                at_ts += ln
            if two:
                lv.tick_end()
            else:
                lv.tick(flush=True)
            hs = lv.frame_hashes()
            got_good += [int(hs[i]) for i, p in enumerate(lv.pictures()) if p.stream == ids[0]]
        lv.tick(flush=True)
        hs = lv.frame_hashes()
        got_good += [int(hs[i]) for i, p in enumerate(lv.pictures()) if p.stream == ids[0]]
    got_good = [x for k3, x in enumerate(got_good) if k3 == 0 or x != got_good[k3 - 1]]
    if got_good != good_want:
        print("case %d: the INTACT live stream beside the damaged ones came out wrong (%d vs %d pictures)" % (c, len(got_good), len(good_want)), flush=True)
        sys.exit(1)
    if os.environ.get("FUZZ_VERBOSE"):
        print("case %d kind %d %dx%d n=%d: batch found %d pictures, one-picture ABI decoded %d" % (c, kind, w, h, n, got, k), flush=True)
print("%d damaged streams decoded without fault or hang (batch, one-picture ABI, live beside an intact stream that came out right) in %.1fs" % (cases, time.time() - t0))
