"""Randomised sweep of the ONE-PICTURE interface's call patterns on the GPU: the same random stream written in random
chunks (a few bytes to several pictures; decode() is only called when the written data ends on a picture boundary, the
way ts.js's PES packets do -- the reference decodes whatever lies behind a picture start code, complete or not), into an
EVICT store that is short enough to evict or an EXPAND store that has to grow, pulled completely or only in part (so
that pictures decoded AHEAD stay queued across writes and evictions), with seeks back to earlier pictures (EXPAND) --
the identical call sequence on the product and on the oracle (checker): every decode()'s return value, cursor and
planes must agree.      python tools/fuzz_abi_chunks.py [cases] [seed]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import build, cabi, synth  # noqa: E402

# FUZZ_LIB: another library with the same 15 functions in the product's place (the reference's own C build, oracle/_ref: is the
# call pattern inside the contract, and does the oracle's store behave like the reference's? -- run on the CPU,
# tests/test_oracle_pin.py: 2300 cases, all agree.  The native build shows malloc's leftovers in macroblocks no picture has
# written yet where the JS / wasm builds show the zeros of a fresh typed array / linear memory: oracle/ref_zeroed_malloc.c)
TESTED = os.environ.get("FUZZ_LIB") or build.LIB_HIP
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def md5(planes):
    h = hashlib.md5()
    for p in planes:
        h.update(p.tobytes())
    return h.hexdigest()


def script(es, offs, mode, cap):
    """the call sequence: ('w', a, b) write es[a:b]; ('d', n) up to n decode() calls (n < 0: until false); ('s', k) seek:
    the cursor onto a bit index observed after the k-th decode of this run"""
    n = len(offs) - 1
    bounds = [int(offs[k]) for k in range(1, n)] + [len(es)]          # written data may be decoded when it ends here
    ops, pos, b = [], 0, 0
    unread_from = 0                                                   # everything before this byte has been consumed for sure
    while pos < len(es):
        # the next decode point: 1 .. 6 pictures on (more with EXPAND: runs for the decode-ahead)
        hop = int(rng.integers(1, 7 if mode == cabi.MODE_EVICT else 14))
        b2 = min(b + hop, len(bounds))
        target = bounds[b2 - 1]
        if mode == cabi.MODE_EVICT and target - unread_from > cap // 2:   # never more unread data than the store holds
            b2 = b + 1
            target = bounds[b2 - 1]
        while pos < target:                                           # ... reached in chunks
            kind = rng.integers(0, 4)
            step = int(rng.integers(1, 48)) if kind == 0 else int(rng.integers(48, 4096)) if kind == 1 else target - pos
            if pos == 0:
                step = max(step, 160)                                 # the sequence header (<= 140 bytes with both matrices) in ONE write: did_write reads it at once (mpeg1.c)
            step = min(step, target - pos, 32768)                     # (the reference's EXPAND store overflows on a write of more than its capacity + what is free:
                                                                      #  buffer.c:48-57 -- outside the contract; the capacities below are >= 64 KB)
            ops.append(("w", pos, pos + step))
            pos += step
        b = b2
        if mode == cabi.MODE_EVICT or rng.integers(0, 3) == 0 or pos == len(es):
            ops.append(("d", -1))
            unread_from = pos
        else:
            ops.append(("d", int(rng.integers(0, hop + 2))))          # leaves pictures buffered (and queued, when decoded ahead)
        if mode == cabi.MODE_EXPAND and rng.integers(0, 6) == 0:
            ops.append(("s", int(rng.integers(0, 1 << 30))))
    ops.append(("d", -1))
    return ops


def run(path, es, ops, mode, cap):
    out, seen = [], []
    with cabi.Mpeg1Decoder(path, cap, mode) as d:
        for op in ops:
            if op[0] == "w":
                d.write(es[op[1]:op[2]])
            elif op[0] == "s":
                if seen:
                    d.index = seen[op[1] % len(seen)]
                    out.append(("seek", d.index))
            else:
                k = 0
                while op[1] < 0 or k < op[1]:
                    got = d.decode()
                    out.append((got, d.index, md5(d.planes()) if got else None))
                    if not got:
                        break
                    seen.append(d.index)
                    k += 1
        ahead = d.ahead_stats() if path == build.LIB_HIP else None
    return out, ahead


bad = served = 0
for c in range(cases):
    w, h = int(rng.integers(1, 23)) * 16 - int(rng.integers(0, 16)), int(rng.integers(1, 19)) * 16 - int(rng.integers(0, 16))
    ov = dict(width=max(w, 2), height=max(h, 2), gop=int(rng.choice([1, 2, 5, 12, 40])), ac_max=int(rng.choice([0, 3, 24, 63])),
              coded_permille=int(rng.choice([50, 400, 950])), f_code_max=int(rng.integers(1, 8)), syntax_quirks=int(rng.choice([0, 0, 1, 2, 3, 5, 7])),
              mv_jitter=int(rng.choice([0, 2, 6])), qscale_lo=int(rng.integers(1, 8)), qscale_hi=int(rng.integers(8, 32)),
              escape_permille=int(rng.choice([0, 20, 300, 1000])), custom_quant=int(rng.integers(0, 2)), quirk_levels=int(rng.integers(0, 2)),
              dc_size_max=int(rng.integers(0, 9)))
    if ov["syntax_quirks"] & 2:
        ov["ac_max"], ov["dc_size_max"] = max(ov["ac_max"], 1), max(ov["dc_size_max"], 2)
    n = int(rng.integers(3, 40))
    try:
        es, offs = synth.generate_config("cfg1_720p", n_frames=n, stream=7000 + c, **ov)
    except RuntimeError as e:
        print("case %d: generator: %s" % (c, e)); continue
    es = np.ascontiguousarray(es, dtype=np.uint8)
    biggest = max(int(offs[k + 1]) - int(offs[k]) for k in range(len(offs) - 2)) if len(offs) > 2 else len(es)
    mode = cabi.MODE_EVICT if rng.integers(0, 2) else cabi.MODE_EXPAND
    cap = int(max(4096, biggest * int(rng.integers(3, 9)))) if mode == cabi.MODE_EVICT else int(rng.choice([65536, 100000, len(es) + 1024]))
    ops = script(es, offs, mode, cap)
    want, _ = run(build.LIB_ORACLE, es, ops, mode, cap)
    got, ahead = run(TESTED, es, ops, mode, cap)
    served += int(ahead[1]) if ahead else 0
    if got != want:
        bad += 1
        k = next((i for i, (a, b_) in enumerate(zip(got, want)) if a != b_), min(len(got), len(want)))
        print("case %d MISMATCH at call %d of %d/%d: got %r want %r | mode %s cap %d n %d %r" % (
            c, k, len(got), len(want), got[k] if k < len(got) else None, want[k] if k < len(want) else None,
            "EVICT" if mode == cabi.MODE_EVICT else "EXPAND", cap, n, ov), flush=True)
    elif os.environ.get("FUZZ_VERBOSE"):
        print("case %d ok: %d calls, mode %s, ahead %r" % (c, len(got), "EVICT" if mode == cabi.MODE_EVICT else "EXPAND", ahead), flush=True)
print("%d cases, %d mismatches; %d pictures served from the decode-ahead" % (cases, bad, served))
sys.exit(1 if bad else 0)
