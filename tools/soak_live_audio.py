"""SOAK of the live AUDIO interface (include/jsmpeg_hip.h part 6): many thousands of ticks over one handle whose streams come and
go.  S slots; a slot's stream plays one of C contents (every generator configuration) from its first frame, LOOPING it -- the
synthesis state runs on across the loops, so the expected samples are the oracle's of the content decoded TWICE in a row: the first
pass for a stream's first loop, the second pass for every later one (a frame looks back 15 sub-blocks, less than one frame) -- 0-2
frames per tick, and now and then leaves; the slot's next stream joins some ticks later with other content ON THE SAME ID (its ring
must read as silence).  Checked every tick: each frame's samples bit for bit, pts, sampling rate, the stream's counters; watched:
the process's resident memory and the device's free memory from the first thousand ticks to the last.

    python tools/soak_live_audio.py [--ticks 40000] [--slots 48] [--out profiles/r06_soak_live_audio.txt]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import build, cabi, mp2, synth  # noqa: E402


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6


def device_free_mb():
    import torch
    return torch.cuda.mem_get_info()[0] / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=40000)
    ap.add_argument("--slots", type=int, default=48)
    ap.add_argument("--seed", type=int, default=6)
    ap.add_argument("--out")
    a = ap.parse_args()
    N = 20
    rng = np.random.default_rng(a.seed)
    oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    contents = []
    for c, name in enumerate(synth.MP2_CONFIGS):
        data, offs = synth.generate_mp2_config(name, N, stream=700 + c)
        b = [int(o) for o in offs] + [len(data)]
        frames = [data[b[k]:b[k + 1]] for k in range(N)]
        want = cabi.decode_mp2_stream(oracle, np.concatenate([data, data]))[0]
        assert len(want) == 2 * N
        thrice = cabi.decode_mp2_stream(oracle, np.concatenate([data, data, data]))[0]
        assert np.array_equal(thrice[2 * N:].view(np.uint32), want[N:].view(np.uint32))       # the premise: every later loop sounds like the second
        contents.append((frames, want.view(np.uint32)))
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)
    say("soak_live_audio: %d ticks, %d slots, %d contents of %d frames (looped), seed %d" % (a.ticks, a.slots, len(contents), N, a.seed))
    frames_checked = mismatches = joins = leaves = 0
    device_free_mb()
    t_start = time.time()
    with mp2.Mp2Live(a.slots, max_frames_per_tick=3, store_bytes=1 << 15) as live:
        slots = [None] * a.slots                    # {id, content, given, checked}
        rejoin = [int(rng.integers(0, 40)) for _ in range(a.slots)]
        marks = {}
        for t in range(a.ticks):
            for i in range(a.slots):
                s = slots[i]
                if s is None:
                    if rejoin[i] > t:
                        continue
                    slots[i] = s = dict(id=live.open(), content=int(rng.integers(0, len(contents))), given=0, checked=0)
                    joins += 1
                elif rng.random() < 1 / 300:
                    info = live.stream_info(s["id"])
                    assert info["frames"] == s["checked"] and info["evictions"] == 0 and info["pending_bytes"] == 0, (info, s["checked"])
                    live.close_stream(s["id"])
                    slots[i] = None
                    rejoin[i] = t + int(rng.integers(1, 30))
                    leaves += 1
                    continue
                fr = contents[s["content"]][0]
                for _ in range(int(rng.choice([0, 1, 1, 1, 2]))):
                    live.write(s["id"], float(s["given"]), fr[s["given"] % N])
                    s["given"] += 1
            n = live.tick()
            by_id = {s["id"]: s for s in slots if s is not None}
            if n:
                pcm = live.read_pcm().view(np.uint32)
                for i, f in enumerate(live.frames()):
                    s = by_id[f["stream"]]
                    k = s["checked"]
                    want = contents[s["content"]][1][k if k < N else N + k % N]
                    if not np.array_equal(pcm[i], want) or f["pts"] != float(k):
                        mismatches += 1
                        if mismatches < 10:
                            say("  MISMATCH tick %d stream %d content %d frame %d (pts %r)" % (t, f["stream"], s["content"], k, f["pts"]))
                    s["checked"] += 1
                    frames_checked += 1
            if t in (1000, a.ticks - 1) or (t and t % 10000 == 0):
                marks[t] = (rss_mb(), device_free_mb())
                say("  tick %6d: %8d frames checked, %d mismatches, %d joins, %d leaves; resident %.1f MB, device free %.1f MB" % (t, frames_checked, mismatches, joins, leaves, *marks[t]))
    dt = time.time() - t_start
    first, last = marks.get(1000), marks.get(a.ticks - 1)
    say("soak_live_audio: %d ticks in %.0f s: %d frames checked against the oracle, %d MISMATCHES; %d streams joined, %d left" % (a.ticks, dt, frames_checked, mismatches, joins, leaves))
    if first and last:
        say("soak_live_audio: resident memory %.1f -> %.1f MB (%+.1f), device free %.1f -> %.1f MB (%+.1f) from tick 1000 to the last" % (first[0], last[0], last[0] - first[0], first[1], last[1], last[1] - first[1]))
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    sys.exit(1 if mismatches else 0)


if __name__ == "__main__":
    main()
