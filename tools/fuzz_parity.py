"""Randomised parity sweep on the GPU: random picture sizes and generator parameters (incl. the unusual-syntax option),
every frame of every stream through the batch interface, the one-picture ABI and the device RGBA stage against the
oracle (checker).  python tools/fuzz_parity.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import batch as jb, build, cabi, hashing, synth  # noqa: E402
from oracle import checkers

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    w, h = int(rng.integers(1, 52)) * 16 - int(rng.integers(0, 16)), int(rng.integers(1, 30)) * 16 - int(rng.integers(0, 16))
    ov = dict(width=max(w, 2), height=max(h, 2), gop=int(rng.choice([1, 2, 3, 5, 9, 12, 15, 40])), ac_max=int(rng.choice([0, 1, 3, 8, 24, 63])),
              qscale_lo=int(rng.integers(1, 8)), qscale_hi=int(rng.integers(8, 32)), escape_permille=int(rng.choice([0, 20, 300, 1000])),
              custom_quant=int(rng.integers(0, 2)), quirk_levels=int(rng.integers(0, 2)), dc_size_max=int(rng.integers(0, 9)),
              coded_permille=int(rng.choice([50, 400, 950])), f_code_max=int(rng.integers(1, 8)), syntax_quirks=int(rng.choice([0, 1, 2, 3, 5, 7])),
              mv_jitter=int(rng.choice([0, 0, 1, 2, 6])))
    if ov["syntax_quirks"] & 2:      # the sweep tells consumed-not-decoded pictures by the repeat they cause: real pictures must differ
        ov["ac_max"], ov["dc_size_max"] = max(ov["ac_max"], 1), max(ov["dc_size_max"], 2)
    n = int(rng.integers(2, 20))
    n_streams = int(rng.integers(1, 4))
    if os.environ.get("FUZZ_VERBOSE"):
        print("case %d: frames=%d streams=%d %r" % (c, n, n_streams, ov), flush=True)
    try:
        gen = [synth.generate_config("cfg1_720p", n_frames=n, stream=1000 * c + s, **ov) for s in range(n_streams)]
        streams = [g[0] for g in gen]
    except RuntimeError as e:
        print("case %d: generator: %s" % (c, e)); continue
    want, want_abi = [], []
    for es in streams:
        frames, _, info = cabi.decode_stream(build.LIB_ORACLE, es, keep="planes")
        want_abi.append(frames)       # one entry per decode() == true: a consumed-not-decoded picture (B / D / f_code 0) repeats the one before
        want.append(frames)           # ... and the batch interface is compared picture by picture where IT says "decoded" (a decoded
                                      # picture may equal the one before too: a P picture without slices shows the picture before last)
    ok = True
    why = []
    with jb.Batch(ov["width"], ov["height"], n_streams, 2 * n_streams * n + 4, sum(len(s) for s in streams) + 4096) as b:
        b.upload(streams)
        got_n = b.decode()
        dev = b.frame_hashes()
        per, seen = {}, {}
        for p, inf in enumerate(b.pictures()):
            k = seen.get(inf.stream, 0)                  # the k-th picture header of its stream = the oracle's k-th decode()
            seen[inf.stream] = k + 1
            if inf.decoded:
                per.setdefault(inf.stream, []).append((p, k))
        for s in range(n_streams):
            distinct = sum(1 for i, f in enumerate(want[s]) if i == 0 or not all(np.array_equal(a, bb) for a, bb in zip(f, want[s][i - 1])))
            if seen.get(s, 0) != len(want[s]) or len(per.get(s, [])) < distinct or \
               any(int(dev[p]) != hashing.frame_hash(*want[s][k]) for p, k in per.get(s, [])):
                ok = False; why.append("batch stream %d" % s)
        p_last, k_last = per[0][-1]
        if not np.array_equal(b.read_rgba(p_last), checkers.oracle_rgba(build.LIB_ORACLE, *want[0][k_last], ov["width"], ov["height"])):
            ok = False; why.append("rgba")
    # the ORDERED reconstruct (one dependency-ordered launch: classes of streams in lockstep, FUZZ_ORDERED=<streams in lockstep>;
    # with JSMPEG_HIP_RECON_CHAINS=1 in the environment: classes of GOP chains): the case's streams eight times over in one batch,
    # every copy's pictures against the oracle
    if os.environ.get("FUZZ_ORDERED"):
        os.environ["JSMPEG_HIP_RECON_ORDER"] = os.environ["FUZZ_ORDERED"]
        try:
            b8 = jb.Batch(ov["width"], ov["height"], 8 * n_streams, 2 * 8 * n_streams * n + 4, 8 * (sum(len(s) for s in streams) + 64 * n_streams) + 4096)
        finally:
            del os.environ["JSMPEG_HIP_RECON_ORDER"]
        with b8:
            b8.upload(streams * 8)
            b8.decode()
            dev8 = b8.frame_hashes()
            info8 = b8.recon_info()
            seen8 = {}
            for p, inf in enumerate(b8.pictures()):
                k = seen8.get(inf.stream, 0)
                seen8[inf.stream] = k + 1
                if inf.decoded and int(dev8[p]) != hashing.frame_hash(*want[inf.stream % n_streams][k]):
                    ok = False; why.append("ordered batch stream %d picture %d (%r)" % (inf.stream, k, info8)); break
    got, _, _ = cabi.decode_stream(build.LIB_HIP, streams[0], keep="planes")
    if len(got) != len(want_abi[0]) or any(not all(np.array_equal(a, bb) for a, bb in zip(x, y)) for x, y in zip(got, want_abi[0])):
        ok = False; why.append("decoder abi")
    # streaming: one write per picture into an EVICT store a few pictures large (how ts.js + the Player drive it)
    big = int(max(np.diff(gen[0][1]))) if n > 1 else len(streams[0])
    got_s, _, _ = cabi.decode_stream(build.LIB_HIP, streams[0], gen[0][1], buffer_size=4 * big + 4096, mode=cabi.MODE_EVICT)
    want_s, _, _ = cabi.decode_stream(build.LIB_ORACLE, streams[0], gen[0][1], buffer_size=4 * big + 4096, mode=cabi.MODE_EVICT)
    if got_s != want_s:
        ok = False; why.append("evict streaming: %d vs %d frames, first diff %s" % (len(got_s), len(want_s), [i for i, (a, bb) in enumerate(zip(got_s, want_s)) if a != bb][:3]))
    # the same streams as MPEG-TS through the device demux, against the ts.js restatement feeding the oracle (a last
    # PES that ends without stuffing stays pending in ts.js: that picture never reaches the decoder, in both)
    tss = [synth.mux_ts(g[0], g[1]) for g in gen]
    want_ts = []
    for ts in tss:
        demuxed, writes = checkers.oracle_ts_demux(build.LIB_ORACLE, ts, 0xE0)
        given = demuxed[:sum(w[2] for w in writes)]
        fr = cabi.decode_stream(build.LIB_ORACLE, given, keep="planes")[0] if len(given) else []
        want_ts.append([hashing.frame_hash(*f) for f in fr])
    with jb.Batch(ov["width"], ov["height"], n_streams, 2 * n_streams * n + 4, sum(len(s) for s in streams) + 4096) as b:
        b.upload_ts(tss)
        b.decode()
        dev_ts = b.frame_hashes()
        per_ts, seen_ts = {}, {}
        for p, inf in enumerate(b.pictures()):
            k = seen_ts.get(inf.stream, 0)
            seen_ts[inf.stream] = k + 1
            if inf.decoded:
                per_ts.setdefault(inf.stream, []).append((int(dev_ts[p]), k))
        for s in range(n_streams):
            if seen_ts.get(s, 0) != len(want_ts[s]) or any(h != want_ts[s][k] for h, k in per_ts.get(s, [])):
                ok = False; why.append("ts path stream %d: %d vs %d pictures" % (s, seen_ts.get(s, 0), len(want_ts[s])))
    if not ok:
        bad += 1
        print("case %d MISMATCH (%s): frames=%d streams=%d params=%r" % (c, "; ".join(why), n, n_streams, ov))
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
