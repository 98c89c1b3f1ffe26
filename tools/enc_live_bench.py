"""On the GPU box: LIVE streams (include/jsmpeg_hip.h part 5) of CODED VIDEO -- the committed encoder-made 1080p GOPs
(tests/golden/enc1080/, tools/enc_content_bench.py's streams), 64 streams, a picture per stream per tick, with the streams'
GOPs NOT aligned: stream s joins at tick s % 12, so every tick carries the intra pictures of about a twelfth of the streams
(what a server of independent streams sees; tools/live_bench.py's streams all start their GOPs together).  A tick lasts as
long as its longest slice's walk: with an intra picture of 0.43-0.52 MB among its pictures that is the 7 ms of the batch path
(profiles/r06k_enc_content.md), not the 1.1 ms of a tick of the generator's P pictures.  ms per tick (host clock, writes
included) and pictures per second; every picture's device hash against the oracle's decoder.
    python tools/enc_live_bench.py [streams] [GOPs per stream] [--aligned] [--json out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import enc_content_bench as ecb  # noqa: E402
from jsmpeg_amd import build, cabi, hashing, live as jl  # noqa: E402


def run(n_streams=64, gops_per_stream=3, aligned=False):
    files = ecb.gop_files(None)
    gops = []
    for k in sorted(files):
        es = np.fromfile(files[k], dtype=np.uint8)
        gops.append(es[:-4])
    END = np.frombuffer(ecb.END, np.uint8)
    distinct = [np.concatenate([gops[(f + k) % len(gops)] for k in range(gops_per_stream)] + [END]) for f in range(len(gops))]

    def writes_of(es):     # one write per picture, as ts.js delivers them: from a picture's first byte (its headers) to the next one's
        at = np.flatnonzero((es[:-3] == 0) & (es[1:-2] == 0) & (es[2:-1] == 1) & (es[3:] == 0))
        cuts = [0]
        for a in at[1:]:
            # a GOP's sequence + group headers travel with ITS first picture: cut at the sequence header in front of the picture code, if any
            seq = a
            back = es[max(0, a - 200):a]
            j = np.flatnonzero((back[:-3] == 0) & (back[1:-2] == 0) & (back[2:-1] == 1) & (back[3:] == 0xB3))
            if j.size:
                seq = max(0, a - 200) + int(j[-1])
            cuts.append(int(seq))
        cuts.append(len(es))
        return [es[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
    writes_d = [writes_of(es) for es in distinct]
    want_d = []
    for es in distinct:
        h = []
        with cabi.Mpeg1Decoder(build.LIB_ORACLE, len(es) + 1024, cabi.MODE_EXPAND) as dec:
            dec.write(es)
            while dec.decode():
                h.append(hashing.frame_hash(*dec.planes()))
        want_d.append(h)
    n_pics = len(writes_d[0])
    assert all(len(w) == n_pics for w in writes_d) and all(len(h) == n_pics for h in want_d)
    delay = [0 if aligned else s % 12 for s in range(n_streams)]
    biggest = max(len(w) for ws in writes_d for w in ws)
    ticks, got, ids = [], {}, {}
    with jl.Live(ecb.W, ecb.H, n_streams, pictures_per_tick=1, store_bytes=max(512 * 1024, 2 * biggest)) as lv:
        for k in range(n_pics + max(delay)):
            t0 = time.perf_counter()
            wrote = i_pics = 0
            for s in range(n_streams):
                j = k - delay[s]
                if j == 0:
                    ids[s] = lv.open()
                    got[ids[s]] = []
                if 0 <= j < n_pics:
                    lv.write(ids[s], writes_d[s % len(distinct)][j], pts=j / 30.0)
                    wrote += 1
                    i_pics += 1 if j % 12 == 0 else 0
            t1 = time.perf_counter()
            n = lv.tick(flush=True)
            t2 = time.perf_counter()
            hs = lv.frame_hashes()
            for i, p in enumerate(lv.pictures()):
                got[p.stream].append(int(hs[i]))
            ticks.append(dict(ms=(t2 - t0) * 1e3, ms_writes=(t1 - t0) * 1e3, pictures=n, written=wrote, intra=i_pics, parts=lv.timings()))
    bad = sum(sum(1 for a, b in zip(got[ids[s]], want_d[s % len(distinct)]) if a != b) + abs(len(got[ids[s]]) - n_pics) for s in range(n_streams))
    full = [t for t in ticks[1:] if t["written"] == n_streams]            # every stream delivers a picture
    med = lambda xs: float(np.median(xs)) if len(xs) else None           # noqa: E731
    with_i = [t for t in full if t["intra"] > 0]
    without = [t for t in full if t["intra"] == 0]
    res = {
        "workload": "%d live 1080p streams of encoder-made content (tests/golden/enc1080/), %d pictures each, a picture per stream per tick, GOPs %s"
                    % (n_streams, n_pics, "aligned (every stream's intra picture in the same tick)" if aligned else "staggered (stream s joins at tick s %% 12)"),
        "ticks": len(ticks), "ticks_with_every_stream": len(full), "pictures": sum(t["pictures"] for t in ticks), "pictures_differing_from_oracle": bad,
        "ms_per_tick_median": med([t["ms"] for t in full]), "ms_per_tick_max": max(t["ms"] for t in full) if full else None,
        "ms_per_tick_with_intra_pictures": med([t["ms"] for t in with_i]), "intra_pictures_per_such_tick": med([t["intra"] for t in with_i]),
        "ms_per_tick_without_intra_pictures": med([t["ms"] for t in without]), "ticks_without_intra_pictures": len(without),
        "ms_writes_median": med([t["ms_writes"] for t in full]),
        "pictures_per_s": sum(t["pictures"] for t in full) / (sum(t["ms"] for t in full) * 1e-3) if full else None,
        "parts_ms_median": {k2: med([t["parts"][k2] for t in full]) for k2 in (full[0]["parts"] if full else {})},
        "bytes_per_write_mean": int(np.mean([len(w) for ws in writes_d for w in ws])), "intra_write_bytes": [int(len(ws[0])) for ws in writes_d],
    }
    for k2, v in list(res.items()):
        if isinstance(v, float):
            res[k2] = round(v, 4)
    res["parts_ms_median"] = {k2: round(v, 4) for k2, v in res["parts_ms_median"].items()}
    if bad:
        raise RuntimeError("PARITY FAILURE: %d live pictures differ from the oracle" % bad)
    return res


if __name__ == "__main__":
    argv = sys.argv[1:]
    out_json = argv[argv.index("--json") + 1] if "--json" in argv else None
    if out_json:
        del argv[argv.index("--json"):argv.index("--json") + 2]
    argv = [a for a in argv if not a.startswith("--")]
    r = run(int(argv[0]) if argv else 64, int(argv[1]) if len(argv) > 1 else 3, aligned="--aligned" in sys.argv)
    print(json.dumps(r))
    if out_json:
        with open(out_json, "w") as f:
            json.dump(r, f, indent=1)
