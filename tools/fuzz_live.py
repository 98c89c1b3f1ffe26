"""Randomised parity sweep of the LIVE interface on the GPU (include/jsmpeg_hip.h part 5): random picture sizes and
generator parameters (incl. the unusual-syntax options: B / D pictures, slices that end mid-row ...), 1-6 streams that join
and leave at random ticks, fed in one of three ways per case --
  flush     : whole pictures per write (0-3 per stream and tick), FLUSH ticks: per stream == the oracle's decoder (EVICT store
              of the same size) fed the same writes with `while (decode());` per tick -- evictions included (small stores);
  pieces    : arbitrary byte pieces (1 byte .. several pictures), ticks without FLUSH + one FLUSH at the end: the pictures of
              the whole stream decoded in one piece;
  ts        : the stream as MPEG-TS in arbitrary byte pieces through jsmpeg_hip_live_write_ts, FLUSH ticks: the pictures of
              what the reference demuxer's restatement delivers, with its pts.
    python tools/fuzz_live.py [cases] [seed]
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import build, cabi, hashing, live as jl, synth  # noqa: E402
from oracle import checkers  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ORACLE = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()


def picture_writes(es, offs):
    n = len(offs) - 1
    return [es[int(offs[k]):(len(es) if k == n - 1 else int(offs[k + 1]))] for k in range(n)]


def oracle_ticks(writes_per_tick, store):
    """hash of every DECODED picture per tick (a consumed-not-decoded picture repeats the planes before it: dropped)"""
    out, last = [], None
    with cabi.Mpeg1Decoder(ORACLE, store, cabi.MODE_EVICT) as dec:
        for ws in writes_per_tick:
            for w in ws:
                dec.write(w)
            tick = []
            while dec.decode():
                h = hashing.frame_hash(*dec.planes())
                if h != last:
                    tick.append(h)
                last = h
            out.append(tick)
    return out


class Ticker:
    """a tick and the collection of its pictures -- or, overlapped, the tick's first half now and its second half (and the
    collection) just before the next tick: everything written in between is written while the pass is on the device"""
    def __init__(self, lv, overlap, collect):
        self.lv, self.overlap, self.collect, self.inflight = lv, overlap, collect, False

    def tick(self, flush):
        if not self.overlap:
            self.lv.tick(flush=flush)
            self.collect()
            return
        self.finish()
        self.lv.tick_begin(flush=flush)
        self.inflight = True

    def finish(self):
        if self.inflight:
            self.lv.tick_end()
            self.collect()
            self.inflight = False


bad = 0
n_overlap = 0
modes = {"flush": 0, "pieces": 0, "ts": 0}
pictures = 0
for c in range(cases):
    w, h = int(rng.integers(1, 40)) * 16 - int(rng.integers(0, 16)), int(rng.integers(1, 24)) * 16 - int(rng.integers(0, 16))
    ov = dict(width=max(w, 2), height=max(h, 2), gop=int(rng.choice([1, 2, 3, 5, 9, 12, 15, 40])), ac_max=int(rng.choice([1, 3, 8, 24, 63])),
              qscale_lo=int(rng.integers(1, 8)), qscale_hi=int(rng.integers(8, 32)), escape_permille=int(rng.choice([0, 20, 300, 1000])),
              custom_quant=int(rng.integers(0, 2)), quirk_levels=int(rng.integers(0, 2)), dc_size_max=int(rng.integers(2, 9)),
              coded_permille=int(rng.choice([50, 400, 950])), f_code_max=int(rng.integers(1, 8)), syntax_quirks=int(rng.choice([0, 0, 1, 2, 3, 5, 7])),
              mv_jitter=int(rng.choice([0, 0, 1, 2, 6])))
    mode = str(rng.choice(["flush", "pieces", "ts"]))
    n_streams, n = int(rng.integers(1, 7)), int(rng.integers(2, 16))
    K = int(rng.integers(1, 6))
    try:
        gen = [synth.generate_config("cfg1_720p", n_frames=n, stream=5000 * c + s, **ov) for s in range(n_streams)]
    except RuntimeError as e:
        print("case %d: generator: %s" % (c, e))
        continue
    if os.environ.get("FUZZ_VERBOSE"):
        print("case %d: %s frames=%d streams=%d K=%d %r" % (c, mode, n, n_streams, K, ov), flush=True)
    modes[mode] += 1
    # every other case: the writes are made WHILE the tick before them is on the device (jsmpeg_hip_live_tick_begin / _end)
    overlap = (c % 2 == 1) if os.environ.get("FUZZ_OVERLAP") is None else os.environ["FUZZ_OVERLAP"] == "1"
    n_overlap += overlap
    noise_rng = np.random.default_rng([seed, c, 99])
    why = []
    biggest = max(len(x) for g in gen for x in picture_writes(g[0], g[1]))
    if mode == "flush":
        # small stores now and then: writes that outrun the ticks throw undecoded bytes away, in both
        store = int(biggest * float(rng.choice([1.2, 2.5, 6.0, 40.0]))) + 64
        join = [int(rng.integers(0, 4)) for _ in range(n_streams)]
        plan = []                                         # per tick {stream: [writes]}
        given = [0] * n_streams
        all_w = [picture_writes(g[0], g[1]) for g in gen]
        noise = {s: noise_rng.integers(0, 256, size=int(min(store, noise_rng.integers(1, 900))), dtype=np.uint8) for s in range(n_streams) if noise_rng.random() < 0.25}
        t = 0
        while any(given[s] < n for s in range(n_streams)) and t < 200:
            row = {}
            for s in range(n_streams):
                if t < join[s] or given[s] >= n:
                    continue
                k = int(rng.choice([0, 1, 1, 1, 2, 3]))
                row[s] = all_w[s][given[s]:given[s] + k]
                given[s] += len(row[s])
                if s in noise and t == join[s]:
                    row[s] = [noise[s]] + row[s]             # bytes that are not video before the stream's first header
            plan.append(row)
            t += 1
        want = [oracle_ticks([row.get(s, []) for row in plan], store) for s in range(n_streams)]
        with jl.Live(ov["width"], ov["height"], n_streams, pictures_per_tick=16, store_bytes=store) as lv:    # (the limit counts picture start codes: B / D pictures too)
            ids, last_live = {}, {}

            def feed(t):
                for s in range(n_streams):
                    if join[s] == t or (t == 0 and join[s] == 0):
                        ids.setdefault(s, lv.open())
                for s, ws in plan[t].items():
                    for x in ws:
                        lv.write(ids[s], x, pts=float(t))
            if overlap and plan:
                feed(0)
            for t, row in enumerate(plan):
                if overlap:
                    lv.tick_begin(flush=True)
                    if t + 1 < len(plan):
                        feed(t + 1)                          # while tick t is on the device
                    lv.tick_end()
                else:
                    feed(t)
                    lv.tick(flush=True)
                hs = lv.frame_hashes()
                per = {}
                for i, p in enumerate(lv.pictures()):
                    # (a decoded picture identical to the one before it -- nothing coded, zero vectors -- is dropped on both sides: the
                    # oracle's list cannot tell it from a consumed-not-decoded picture's repeat)
                    if int(hs[i]) != last_live.get(p.stream):
                        per.setdefault(p.stream, []).append(int(hs[i]))
                    last_live[p.stream] = int(hs[i])
                pictures += len(hs)
                for s, i in ids.items():
                    if join[s] > t:
                        continue                             # (opened ahead, beside tick t)
                    if per.get(i, []) != want[s][t]:
                        why.append("tick %d stream %d: %d vs %d pictures" % (t, s, len(per.get(i, [])), len(want[s][t])))
                        if os.environ.get("FUZZ_VERBOSE"):
                            whole = [hashing.frame_hash(*f) for f in cabi.decode_stream(ORACLE, gen[s][0], keep="planes")[0]]
                            name = lambda hh: whole.index(hh) if hh in whole else "?"
                            info = lv.stream_info(i)
                            print("   stream %d (id %d, joined %d): store %d, writes per tick %r" % (s, i, join[s], store, [[len(x) for x in r.get(s, [])] for r in plan[:t + 1]]))
                            print("   live this tick: pictures %r; oracle: %r; live evictions %d pending %d pictures so far %d; oracle per tick so far %r"
                                  % ([name(x) for x in per.get(i, [])], [name(x) for x in want[s][t]], info.evictions, info.pending_bytes, info.pictures, [[name(x) for x in tk] for tk in want[s][:t + 1]]))
                if why:
                    break
    elif mode == "pieces":
        want = []
        for g in gen:
            frames, _, _ = cabi.decode_stream(ORACLE, g[0], keep="planes")
            hs, last = [], None
            for f in frames:
                x = hashing.frame_hash(*f)
                if x != last:
                    hs.append(x)
                last = x
            want.append(hs)
        got = [[] for _ in range(n_streams)]
        with jl.Live(ov["width"], ov["height"], n_streams, pictures_per_tick=K, store_bytes=2 * max(len(g[0]) for g in gen) + 4096) as lv:
            ids = [lv.open() for _ in range(n_streams)]
            at = [0] * n_streams
            mean = max(16, biggest // 2)

            def collect():
                global pictures
                hs = lv.frame_hashes()
                for i, p in enumerate(lv.pictures()):
                    got[ids.index(p.stream)].append(int(hs[i]))
                pictures += len(hs)
            tk = Ticker(lv, overlap, collect)
            while any(at[s] < len(gen[s][0]) for s in range(n_streams)):
                for s in range(n_streams):
                    k = min(len(gen[s][0]) - at[s], int(rng.choice([1, 2, 3, 4, 7, mean // 3, mean, mean, 3 * mean, 9 * mean])))
                    if k > 0:
                        lv.write(ids[s], gen[s][0][at[s]:at[s] + k])
                        at[s] += k
                if rng.random() < 0.7:
                    tk.tick(False)
            # drain: a tick may decode nothing and still move on (K picture START CODES per tick: a B / D picture uses one up)
            for fl in (False, True):
                before = None
                for _ in range(8 * n + 8):
                    tk.tick(fl)
                    tk.finish()
                    now = [lv.stream_info(i).pending_bytes for i in ids]
                    if now == before:
                        break
                    before = now
        for s in range(n_streams):
            got[s] = [x for k, x in enumerate(got[s]) if k == 0 or x != got[s][k - 1]]
            if got[s] != want[s]:
                why.append("stream %d: %d vs %d pictures, first diff %s" % (s, len(got[s]), len(want[s]), [i for i, (a, b) in enumerate(zip(got[s], want[s])) if a != b][:3]))
    else:
        tss = [synth.mux_ts(g[0], g[1]) for g in gen]
        got = [[] for _ in range(n_streams)]
        rounds = [[] for _ in range(n_streams)]             # per stream: the piece it was handed in every round (0: none)
        with jl.Live(ov["width"], ov["height"], n_streams, pictures_per_tick=16, store_bytes=2 * max(len(g[0]) for g in gen) + 4096) as lv:
            ids = [lv.open() for _ in range(n_streams)]
            at = [0] * n_streams

            def collect_ts():
                global pictures
                hs = lv.frame_hashes()
                for i, p in enumerate(lv.pictures()):
                    got[ids.index(p.stream)].append(int(hs[i]))
                pictures += len(hs)
            tk = Ticker(lv, overlap, collect_ts)
            while any(at[s] < len(tss[s]) for s in range(n_streams)):
                for s in range(n_streams):
                    k = min(len(tss[s]) - at[s], int(rng.choice([1, 50, 187, 188, 189, 1000, 5000])))
                    rounds[s].append(k)
                    if k > 0:
                        lv.write_ts(ids[s], tss[s][at[s]:at[s] + k])
                        at[s] += k
                tk.tick(True)
            tk.finish()
            # (a tick looks at 16 picture start codes per stream: a piece of 5000 bytes can bring more of these tiny pictures)
            before = None
            for _ in range(8 * n + 8):
                now = [lv.stream_info(i).pending_bytes for i in ids]
                if now == before or not any(now):
                    break
                before = now
                lv.tick(flush=True)
                hs = lv.frame_hashes()
                for i, p in enumerate(lv.pictures()):
                    got[ids.index(p.stream)].append(int(hs[i]))
                pictures += len(hs)
        for s in range(n_streams):
            # the oracle's decoder fed what the reference demuxer's restatement delivers for the same pieces, `while (decode());`
            # at the same points: after the writes each round completed
            sizes = [k for k in rounds[s] if k]
            demuxed, writes = checkers.oracle_ts_demux(ORACLE, tss[s], 0xE0, sizes)
            done_after, cum = [], 0
            for k in rounds[s]:
                if k:
                    cum += k
                    n_w = len(checkers.oracle_ts_demux(ORACLE, tss[s][:cum], 0xE0, [x for x in sizes[:len([y for y in rounds[s][:len(done_after) + 1] if y])]])[1])
                else:
                    n_w = done_after[-1] if done_after else 0
                done_after.append(n_w)
            groups, prev = [], 0
            for n_w in done_after:
                groups.append([demuxed[o:o + ln] for _, o, ln in writes[prev:n_w]])
                prev = n_w
            want_s = [x for tick in oracle_ticks(groups, 2 * max(len(g[0]) for g in gen) + 4096) for x in tick]
            got[s] = [x for k, x in enumerate(got[s]) if k == 0 or x != got[s][k - 1]]
            if got[s] != want_s:
                why.append("ts stream %d: %d vs %d pictures" % (s, len(got[s]), len(want_s)))
    if why:
        bad += 1
        print("case %d MISMATCH [%s] (%s): frames=%d streams=%d K=%d params=%r" % (c, mode, "; ".join(why[:4]), n, n_streams, K, ov), flush=True)
print("%d cases (%s; %d of them written beside ticks in flight), %d live pictures compared, %d mismatches" % (cases, ", ".join("%s %d" % kv for kv in modes.items()), n_overlap, pictures, bad))
sys.exit(1 if bad else 0)
