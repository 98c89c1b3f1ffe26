"""SOAK of the live-stream interface (include/jsmpeg_hip.h part 5): many thousands of ticks over a handle whose streams come
and go -- the long run a live server is, which the tests' dozens of ticks are not.  S slots; a slot's stream plays one of C
contents from its first picture, LOOPING it (every loop starts with the sequence header and an intra picture, so picture k
of every loop is picture k of the first: the expected hashes are the oracle's of one loop), 0-2 pictures per tick, and now
and then leaves; the slot's next stream joins some ticks later with other content.  Every third tick is made in two halves
with the next tick's writes between them.  Checked every tick: each picture's device hash against the oracle's for that
content and position, pts, the stream's counters; watched: the process's resident memory and the device's free memory from
the first thousand ticks to the last (a leak of anything per tick, per write or per stream shows there).

    python tools/soak_live.py [--ticks 30000] [--slots 48] [--out profiles/r06_soak_live.txt]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import build, cabi, hashing, live as jl, synth  # noqa: E402


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6


def device_free_mb():
    import torch                                     # (the package has imported it already: the library binds the HIP runtime torch ships)
    return torch.cuda.mem_get_info()[0] / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=30000)
    ap.add_argument("--slots", type=int, default=48)
    ap.add_argument("--contents", type=int, default=10)
    ap.add_argument("--seed", type=int, default=6)
    ap.add_argument("--out")
    a = ap.parse_args()
    W, H, N = 640, 368, 36
    rng = np.random.default_rng(a.seed)
    oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    contents = []
    for c in range(a.contents):
        es, offs = synth.generate_config("cfg1_720p", n_frames=N, stream=900 + c, width=W, height=H)
        n = len(offs) - 1
        writes = [es[int(offs[k]):(len(es) if k == n - 1 else int(offs[k + 1]))] for k in range(n)]
        frames, _, _ = cabi.decode_stream(oracle, es, keep="planes")
        want = [hashing.frame_hash(*f) for f in frames]
        # (the loop's premise: looping the stream through the oracle gives the same pictures again)
        twice, _, _ = cabi.decode_stream(oracle, np.concatenate([es, es]), keep="planes")
        assert [hashing.frame_hash(*f) for f in twice] == want + want
        contents.append((writes, want))
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)
    say("soak_live: %d ticks, %d slots of %dx%d, %d contents of %d pictures (looped), seed %d" % (a.ticks, a.slots, W, H, a.contents, N, a.seed))
    store = 4 * max(len(w) for ws, _ in contents for w in ws)
    pictures = mismatches = joins = leaves = halves = 0
    device_free_mb()                                 # (torch's own context on the device: made now, not between the marks)
    t_start = time.time()
    with jl.Live(W, H, a.slots, pictures_per_tick=4, store_bytes=store) as lv:
        slots = [None] * a.slots                    # {id, content, given, checked, wait}
        rejoin = [int(rng.integers(0, 40)) for _ in range(a.slots)]
        marks = {}

        def feed(t, in_flight):
            nonlocal joins, leaves
            for i in range(a.slots):
                s = slots[i]
                if s is None:
                    if rejoin[i] <= t:
                        slots[i] = s = dict(id=lv.open(), content=int(rng.integers(0, a.contents)), given=0, checked=0)
                        joins += 1
                    else:
                        continue
                elif rng.random() < 1 / 300:
                    if not in_flight:                # (beside a tick its pictures are not counted yet)
                        info = lv.stream_info(s["id"])
                        assert info.pictures == s["checked"] and info.evictions == 0, (info.pictures, s["checked"], info.evictions)
                    lv.close_stream(s["id"])
                    slots[i] = None
                    rejoin[i] = t + int(rng.integers(1, 30))
                    leaves += 1
                    continue
                writes = contents[s["content"]][0]
                for _ in range(int(rng.choice([0, 1, 1, 1, 2]))):
                    lv.write(s["id"], writes[s["given"] % N], pts=float(s["given"]))
                    s["given"] += 1

        def check(t, by_id):
            nonlocal pictures, mismatches
            hs = lv.frame_hashes()
            for i, p in enumerate(lv.pictures()):
                s = by_id[p.stream]
                want = contents[s["content"]][1][s["checked"] % N]
                if int(hs[i]) != want or p.pts != float(s["checked"]):
                    mismatches += 1
                    if mismatches < 10:
                        say("  MISMATCH tick %d stream %d content %d picture %d (pts %r)" % (t, p.stream, s["content"], s["checked"], p.pts))
                s["checked"] += 1
                pictures += 1
        feed(0, False)
        for t in range(a.ticks):
            two = t % 3 == 2
            if two:
                lv.tick_begin(flush=True)
                in_pass = {s["id"]: s for s in slots if s is not None}      # (a stream may leave, and its id be taken, beside the tick)
                feed(t + 1, True)
                lv.tick_end()
                halves += 1
            else:
                lv.tick(flush=True)
                in_pass = {s["id"]: s for s in slots if s is not None}
            check(t, in_pass)
            if not two:
                feed(t + 1, False)
            if t in (1000, a.ticks - 1) or (t and t % 5000 == 0):
                marks[t] = (rss_mb(), device_free_mb())
                say("  tick %6d: %8d pictures checked, %d mismatches, %d joins, %d leaves; resident %.1f MB, device free %.1f MB" % (t, pictures, mismatches, joins, leaves, *marks[t]))
        pend = [lv.stream_info(s["id"]).pending_bytes for s in slots if s is not None]
    dt = time.time() - t_start
    first, last = marks.get(1000), marks.get(a.ticks - 1)
    say("soak_live: %d ticks (%d in two halves with writes between them) in %.0f s: %d pictures checked against the oracle, %d MISMATCHES; %d streams joined, %d left; "
        "pending bytes at the end: %d" % (a.ticks, halves, dt, pictures, mismatches, joins, leaves, sum(pend)))
    if first and last:
        say("soak_live: resident memory %.1f -> %.1f MB (%+.1f), device free %.1f -> %.1f MB (%+.1f) from tick 1000 to the last" % (first[0], last[0], last[0] - first[0], first[1], last[1], last[1] - first[1]))
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    sys.exit(1 if mismatches else 0)


if __name__ == "__main__":
    main()
