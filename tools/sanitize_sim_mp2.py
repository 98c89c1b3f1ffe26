"""The MP2 kernels' device functions under AddressSanitizer + UBSan -- on the CPU, where sanitizers exist (GPU ASan is not available
on the pool).  tests/sim/sim_mp2.cpp compiles the kernels' own header (mp2_dev.h: the frame walk, the five side-information phases,
sample read + requantisation, matrixing, windowing -- incl. the LIVE placement of C ABI part 6) with g++; tests/sim/sim_mp2_main.cpp
makes a program of it.  This tool builds that program with -fsanitize=address,undefined and feeds it
  * every golden MP2 fixture's stream as it is: one batch pass, and live ticks with the bytes in pseudo-random pieces (1, 2 and 5
    frame places per tick) -- all four must decode the fixture's number of frames to the same samples, and
  * DAMAGED copies -- flipped bits, overwritten runs, truncations, dropped bytes, headers dropped in -- which is where a frame's
    allocation promises more bits than it has, scalefactor indices and sample codes take values no encoder writes, and a parser
    reads or writes past its buffers if it ever does: on the GPU such an access is silent, here it stops the program.

    python tools/sanitize_sim_mp2.py [--damaged 60] [--seed 1] [--out profiles/rNN_sanitize_sim_mp2.txt]
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import synth  # noqa: E402

BIN = os.path.join(ROOT, "tests", "sim", "_asan", "sim_mp2_main_asan")


def build():
    src = os.path.join(ROOT, "tests", "sim", "sim_mp2_main.cpp")
    csrc = os.path.join(ROOT, "jsmpeg_amd", "csrc")
    deps = [src, os.path.join(ROOT, "tests", "sim", "sim_mp2.cpp")] + glob.glob(os.path.join(csrc, "mp2_*.h"))
    if os.path.exists(BIN) and all(os.path.getmtime(d) <= os.path.getmtime(BIN) for d in deps):
        return BIN
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off",
                           "-fno-omit-frame-pointer", "-Wno-unknown-pragmas", "-I", csrc, "-o", BIN, src])
    return BIN


def damage(data, rng):
    b = data.copy()
    n = len(b)
    kind = int(rng.integers(0, 6))
    if kind == 0:
        k = int(rng.integers(1, 40))
        for p in rng.integers(0, n, size=k):
            b[p] ^= 1 << int(rng.integers(0, 8))
        return b, "%d bits flipped" % k
    if kind == 1:
        at, ln = int(rng.integers(0, n)), int(rng.integers(1, 300))
        b[at:at + ln] = rng.integers(0, 256, size=len(b[at:at + ln]), dtype=np.uint8)
        return b, "%d random bytes at %d" % (ln, at)
    if kind == 2:
        cut = int(rng.integers(4, n))
        return b[:cut], "cut at %d of %d" % (cut, n)
    if kind == 3:
        at, ln = int(rng.integers(0, n)), int(rng.integers(1, 200))
        return np.concatenate([b[:at], b[at + ln:]]), "%d bytes dropped at %d" % (ln, at)
    if kind == 4:
        k = int(rng.integers(1, 6))
        for p in rng.integers(0, max(1, n - 4), size=k):                 # a header of any bit rate / sampling frequency / mode, CRC or not
            b[p:p + 4] = [0xff, 0xfc | int(rng.integers(0, 2)), int(rng.integers(0, 256)), int(rng.integers(0, 256))]
        return b, "%d frame headers dropped in" % k
    at, ln = int(rng.integers(0, n)), int(rng.integers(1, 2000))
    b[at:at + ln] = int(rng.choice([0x00, 0xFF]))
    return b, "%d bytes of one value at %d" % (ln, at)


def run(binary, data, args, td):
    path = os.path.join(td, "s.mp2")
    data.tofile(path)
    r = subprocess.run([binary, path] + args, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    return r.returncode, r.stdout.strip(), r.stderr.strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--damaged", type=int, default=60, help="damaged copies per fixture")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out")
    a = ap.parse_args()
    binary = build()
    rng = np.random.default_rng(a.seed)
    lines, bad, runs = [], 0, 0
    t0 = time.time()

    def say(s):
        print(s, flush=True)
        lines.append(s)
    say("sanitize_sim_mp2: %s (g++ -fsanitize=address,undefined), %d damaged copies per fixture, seed %d" % (os.path.relpath(binary, ROOT), a.damaged, a.seed))
    modes = [["batch"], ["live", "1", "11"], ["live", "2", "12"], ["live", "5", "13"]]
    with tempfile.TemporaryDirectory() as td:
        for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mp2_*.json"))):
            name = os.path.basename(path)[4:-5]
            fx = json.load(open(path))
            data, _ = synth.generate_mp2_config(fx["config"], fx["n_frames"], **fx["overrides"])
            outs = []
            for m in modes:
                rc, out, err = run(binary, data, m, td)
                runs += 1
                outs.append(out)
                if rc != 0 or not out.startswith("%d frames" % fx["n_frames"]):
                    bad += 1
                    say("  FAIL %s clean (%s): rc %d, %s\n%s" % (name, " ".join(m), rc, out, err[-1500:]))
            if len(set(outs)) != 1:
                bad += 1
                say("  FAIL %s: the batch pass and the live ticks do not give the same samples: %r" % (name, outs))
            fails = 0
            for k in range(a.damaged):
                b, what = damage(data, rng)
                m = modes[k % len(modes)] if k % len(modes) == 0 else ["live", modes[k % len(modes)][1], str(int(rng.integers(1, 1 << 30)))]
                rc, out, err = run(binary, b, m, td)
                runs += 1
                if rc != 0:
                    fails += 1
                    bad += 1
                    keep = os.path.join(ROOT, "gpurun_out", "sanitize_mp2_%s_%d.mp2" % (name, k))
                    os.makedirs(os.path.dirname(keep), exist_ok=True)
                    b.tofile(keep)
                    say("  FAIL %s damaged copy %d (%s; %s): rc %d, %s -> %s\n%s" % (name, k, what, " ".join(m), rc, out, os.path.relpath(keep, ROOT), err[-1500:]))
            say("  %-22s %3d frames: batch + live x 3 ok and alike, %d damaged copies, %d stopped by a sanitizer" % (name, fx["n_frames"], a.damaged, fails))
    say("sanitize_sim_mp2: %d runs in %.0f s, %d stopped by a sanitizer or with the wrong frame count" % (runs, time.time() - t0, bad))
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
