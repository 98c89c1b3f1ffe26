"""Randomised parity sweep of the LIVE AUDIO interface on the GPU (include/jsmpeg_hip.h part 6): random generator parameters
(every sampling frequency, bit rates / modes / CRC / padding changing from frame to frame, forbidden-but-decodable codes,
sparse and dense allocations), 1-6 streams of ONE handle that join and leave at random ticks (ids are reused: a new stream on
an old id must start from a silent synthesis state), 1-5 frame places per stream and tick, fed in one of three ways per case --
  frames : whole frames per write (0-4 per stream and tick; sometimes a piece of noise in between), small stores: per stream and
           tick == the oracle's decoder (EVICT store of the same size) given the same writes with `while (decode());` per tick --
           evictions, stalls at noise and their evacuation included;
  pieces : arbitrary byte pieces (1 byte .. several frames): the frames of the whole stream decoded in one piece;
  ts     : the stream as MPEG-TS (two or three frames per PES) in arbitrary byte pieces through jsmpeg_hip_mp2_live_write_ts:
           the frames of the whole stream, each with its PES's pts.
    python tools/fuzz_live_audio.py [cases] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from jsmpeg_amd import build, cabi, mp2, synth  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ORACLE = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()


def same(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def make_stream(quirks_ok):
    kw = dict(sample_rate_index=int(rng.integers(0, 3)), bitrate_index=int(rng.integers(1, 15)), mode=int(rng.integers(0, 4)),
              crc=int(rng.integers(0, 2)), vary=int(rng.random() < 0.5), quirks=int(quirks_ok and rng.random() < 0.3),
              alloc_permille=int(rng.choice([150, 500, 800, 1000])), sf_lo=int(rng.choice([8, 12, 30])), sf_hi=62,
              seed=int(rng.integers(1, 2 ** 31 - 1)))
    data, offs = synth.generate_mp2(int(rng.integers(2, 25)), **kw)
    return data, [int(o) for o in offs] + [len(data)]


class Member:
    """one stream of the case: its content, where the feeding stands, what came out, and -- for `frames` -- the oracle's decoder"""
    def __init__(self, mode, store):
        self.mode = mode
        self.data, self.b = make_stream(quirks_ok=mode != "pieces")      # (a frame that promises more bits than it has reads into what
        self.at = 0                                                       #  FOLLOWS it: in pieces that may not have arrived -- outside the contract)
        self.got, self.pts = [], []
        self.id = None
        self.join = int(rng.integers(0, 6))
        self.dec = cabi.Mp2Decoder(ORACLE, store, cabi.MODE_EVICT) if mode == "frames" else None
        self.want_tick = []
        if mode == "ts":
            from ts_craft import Muxer
            m, k, n = Muxer(), 0, len(self.b) - 1
            self.pes_pts = []
            while k < n:
                hi = min(n, k + int(rng.integers(2, 4)))
                pts = 90000 + 2351 * k
                m.pes(0x101, 0xC0, self.data[self.b[k]:self.b[hi]].tobytes(), pts=pts, with_length=True)
                self.pes_pts += [pts / 90000.0] * (hi - k)
                if rng.random() < 0.3:
                    m.packet(0x1fff, b"")
                k = hi
            self.ts = m.bytes()

    def done(self):
        return self.at >= (len(self.ts) if self.mode == "ts" else len(self.b) - 1 if self.mode == "frames" else len(self.data))


STATS = {"frames": 0, "pieces": 0, "ts": 0, "decoded": 0, "ticks": 0, "evictions": 0, "stalls": 0, "reused_ids": 0, "streams": 0}


def run_case(case):
    mode = ["frames", "pieces", "ts"][int(rng.integers(0, 3))]
    n = int(rng.integers(1, 7))
    cap = int(rng.integers(1, 6)) if mode != "frames" else 24
    store = int(rng.choice([3000, 6000, 20000])) if mode == "frames" else 1 << 17
    members = [Member(mode, store) for _ in range(n)]
    STATS[mode] += 1; STATS["streams"] += n
    used_ids = set()
    slots = max(1, n - int(rng.integers(0, 2)))             # fewer ids than streams now and then: a stream waits for one that leaves
    waiting = list(members)
    active = []
    with mp2.Mp2Live(slots, max_frames_per_tick=cap, store_bytes=store) as live:
        for tick in range(4000):
            for m in list(waiting):
                if tick >= m.join and len(active) < slots:
                    m.id = live.open()
                    STATS["reused_ids"] += m.id in used_ids
                    used_ids.add(m.id)
                    waiting.remove(m); active.append(m)
            for m in active:
                if m.done():
                    continue
                if mode == "frames":
                    ws = []
                    for _ in range(int(rng.integers(0, 3))):
                        k = min(int(rng.integers(1, 5)), len(m.b) - 1 - m.at)
                        while k > 1 and m.b[m.at + k] - m.b[m.at] > store:
                            k -= 1
                        if k < 1 or m.b[m.at + k] - m.b[m.at] > store:
                            break
                        ws.append(m.data[m.b[m.at]:m.b[m.at + k]]); m.at += k
                        if rng.random() < 0.04:
                            ws.append(rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8))   # noise: a stall until the store overflows
                    for w in ws:
                        live.write(m.id, float(tick), w)
                        m.dec.write(w)
                elif mode == "pieces":
                    if rng.random() < 0.7:
                        k = int(rng.choice([1, 3, 50, 333, 1000, 4000]))
                        live.write(m.id, 0.0, m.data[m.at:m.at + k]); m.at += k
                else:
                    k = int(rng.choice([1, 50, 188, 500, 3000]))
                    live.write_ts(m.id, m.ts[m.at:m.at + k]); m.at += k
            cnt = live.tick()
            STATS["ticks"] += 1; STATS["decoded"] += cnt
            frames = live.frames()
            pcm = live.read_pcm()
            by_id = {m.id: m for m in active}
            this_tick = {m.id: [] for m in active}
            for i in range(cnt):
                m = by_id[frames[i]["stream"]]
                this_tick[m.id].append(pcm[i]); m.got.append(pcm[i]); m.pts.append(frames[i]["pts"])
            for m in active:
                if len(this_tick[m.id]) > cap:
                    return "stream took %d frames in a tick of %d places" % (len(this_tick[m.id]), cap)
                if mode == "frames":
                    want = []
                    while m.dec.decode():
                        want.append(np.stack(m.dec.channels()))
                    if len(want) != len(this_tick[m.id]) or any(not same(a, b) for a, b in zip(want, this_tick[m.id])):
                        return "tick %d stream %d: %d frames, the oracle given the same writes %d (or samples differ)" % (tick, m.id, len(this_tick[m.id]), len(want))
            for m in list(active):
                if m.done() and not this_tick[m.id]:                  # everything fed and a tick that brought nothing more: the stream leaves
                    info = live.stream_info(m.id)
                    STATS["evictions"] += info["evictions"]; STATS["stalls"] += info["stalled"]
                    live.close_stream(m.id)
                    active.remove(m)
            if not active and not waiting:
                break
        else:
            return "did not finish"
    for m in members:
        if m.dec:
            m.dec.close()
        if mode in ("pieces", "ts"):
            want = cabi.decode_mp2_stream(ORACLE, m.data)[0]
            if len(m.got) != len(want) or not same(np.array(m.got).reshape(-1, 2, 1152), want):
                return "%s: %d frames, the whole stream %d (or samples differ)" % (mode, len(m.got), len(want))
        if mode == "ts" and any(abs(a - b) > 1e-9 for a, b in zip(m.pts, m.pes_pts)):
            return "ts: pts differ"
    return None


bad = 0
for case in range(cases):
    err = run_case(case)
    if err:
        bad += 1
        print("case %d (seed %d): %s" % (case, seed, err))
print("fuzz_live_audio: %d cases, seed %d, %d mismatches  (%d whole-frame / %d pieces / %d TS cases, %d streams, %d on a reused id, %d ticks, %d frames decoded, %d evictions, %d streams left stalled)"
      % (cases, seed, bad, STATS["frames"], STATS["pieces"], STATS["ts"], STATS["streams"], STATS["reused_ids"], STATS["ticks"], STATS["decoded"], STATS["evictions"], STATS["stalls"]))
sys.exit(1 if bad else 0)
