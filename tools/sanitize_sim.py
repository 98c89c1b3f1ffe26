"""The device functions under AddressSanitizer + UBSan -- on the CPU, where sanitizers exist (GPU ASan is not available on the
pool).  tests/sim/sim_decode.cpp compiles the kernels' own headers (index_tables.h, slice_parse.h, recon_block.h, recon_plan.h)
with g++ and runs them over buffers sized exactly like the device's (the same pads, the same token capacity per byte);
tests/sim/sim_main.cpp makes a program of it.  This tool builds that program with -fsanitize=address,undefined and feeds it
  * every golden fixture's stream as it is (both forms of the parse's ring service), and
  * DAMAGED copies of them -- flipped bits, overwritten runs, truncations, start codes dropped in or knocked out, slice
    rows renumbered -- which is where a parser reads or writes past its buffers if it ever does: on the GPU such an access
    is silent, here it stops the program.
A clean stream must also give the pictures count the fixture records.  (-fno-sanitize=shift-base: `mv << 1` of a negative
vector is the reference's own arithmetic, mpeg1.c:1149-1204, and what the hardware does.)

    python tools/sanitize_sim.py [--damaged 40] [--seed 1] [--max-pixels 414720] [--out profiles/rNN_sanitize_sim.txt]
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

BIN = os.path.join(ROOT, "tests", "sim", "_asan", "sim_main_asan")


def build():
    src = os.path.join(ROOT, "tests", "sim", "sim_main.cpp")
    csrc = os.path.join(ROOT, "jsmpeg_amd", "csrc")
    deps = [src, os.path.join(ROOT, "tests", "sim", "sim_decode.cpp")] + glob.glob(os.path.join(csrc, "*.h"))
    if os.path.exists(BIN) and all(os.path.getmtime(d) <= os.path.getmtime(BIN) for d in deps):
        return BIN
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize=shift-base", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-Wno-unknown-pragmas", "-I", csrc, "-o", BIN, src])
    return BIN


def damage(es, rng):
    """one damaged copy of a stream (a uint8 array); what was done, for the report"""
    b = es.copy()
    kind = int(rng.integers(0, 7))
    n = len(b)
    if kind == 0:
        k = int(rng.integers(1, 40))
        for p in rng.integers(0, n, size=k):
            b[p] ^= 1 << int(rng.integers(0, 8))
        return b, "%d bits flipped" % k
    if kind == 1:
        at, ln = int(rng.integers(0, n)), int(rng.integers(1, 400))
        b[at:at + ln] = rng.integers(0, 256, size=len(b[at:at + ln]), dtype=np.uint8)
        return b, "%d random bytes at %d" % (ln, at)
    if kind == 2:
        cut = int(rng.integers(12, n))
        return b[:cut], "cut at %d of %d" % (cut, n)
    if kind == 3:
        k = int(rng.integers(1, 6))
        for p in rng.integers(0, max(1, n - 4), size=k):
            b[p:p + 4] = [0, 0, 1, int(rng.choice([0x00, 0x01, 0x05, 0xAF, 0xB3, 0xB8, 0xB7, 0xB2, 0xFF]))]
        return b, "%d start codes dropped in" % k
    if kind == 4:
        pos = [i for i in range(n - 3) if b[i] == 0 and b[i + 1] == 0 and b[i + 2] == 1]
        k = min(len(pos), int(rng.integers(1, 5)))
        for p in rng.choice(pos, size=k, replace=False) if k else []:
            b[p + 2] = int(rng.integers(2, 256))
        return b, "%d start codes knocked out" % k
    if kind == 5:
        pos = [i for i in range(n - 3) if b[i] == 0 and b[i + 1] == 0 and b[i + 2] == 1 and 1 <= b[i + 3] <= 0xAF]
        k = min(len(pos), int(rng.integers(1, 8)))
        for p in rng.choice(pos, size=k, replace=False) if k else []:
            b[p + 3] = int(rng.integers(1, 0xB0))
        return b, "%d slices renumbered" % k
    at, ln = int(rng.integers(0, n)), int(rng.integers(1, 3000))
    b[at:at + ln] = int(rng.choice([0x00, 0xFF]))
    return b, "%d bytes of one value at %d" % (ln, at)


def run(binary, es, w, h, frames, split, td):
    path = os.path.join(td, "s.m1v")
    es.tofile(path)
    r = subprocess.run([binary, path, str(w), str(h), str(frames), str(split)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    return r.returncode, r.stdout.strip(), r.stderr.strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--damaged", type=int, default=40, help="damaged copies per fixture")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-pixels", type=int, default=720 * 576, help="fixtures larger than this run clean only")
    ap.add_argument("--only", help="substring of the fixture names to run")
    ap.add_argument("--out")
    a = ap.parse_args()
    binary = build()
    rng = np.random.default_rng(a.seed)
    lines, bad, runs = [], 0, 0
    t0 = time.time()

    def say(s):
        print(s, flush=True)
        lines.append(s)
    say("sanitize_sim: %s (g++ -fsanitize=address,undefined -fno-sanitize=shift-base), %d damaged copies per fixture, seed %d" % (os.path.relpath(binary, ROOT), a.damaged, a.seed))
    with tempfile.TemporaryDirectory() as td:
        for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "frames_*.json"))):
            name = os.path.basename(path)[7:-5]
            if a.only and a.only not in name:
                continue
            fx = json.load(open(path))
            if "es_file" in fx:
                es = np.fromfile(os.path.join(ROOT, "tests", "golden", fx["es_file"]), dtype=np.uint8)
            else:
                es, _ = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
            w, h, n = fx["info"]["width"], fx["info"]["height"], fx["n_frames"]
            for split in (0, 1):
                rc, out, err = run(binary, es, w, h, n + 2, split, td)
                runs += 1
                if rc != 0 or not out.startswith("%d pictures" % len(fx["frame_md5"])):
                    bad += 1
                    say("  FAIL %s clean (ring service %d): rc %d, %s\n%s" % (name, split, rc, out, err[-1500:]))
            n_dam = a.damaged if w * h <= a.max_pixels else 0
            fails = 0
            for k in range(n_dam):
                b, what = damage(es, rng)
                rc, out, err = run(binary, b, w, h, n + 8, k & 1, td)
                runs += 1
                if rc != 0:
                    fails += 1
                    bad += 1
                    keep = os.path.join(ROOT, "gpurun_out", "sanitize_%s_%d.m1v" % (name, k))
                    os.makedirs(os.path.dirname(keep), exist_ok=True)
                    b.tofile(keep)
                    say("  FAIL %s damaged copy %d (%s): rc %d, %s -> %s\n%s" % (name, k, what, rc, out, os.path.relpath(keep, ROOT), err[-1500:]))
            say("  %-34s %4dx%-4d %3d pictures: clean x 2 ok, %d damaged copies, %d stopped by a sanitizer" % (name, w, h, n, n_dam, fails))
    say("sanitize_sim: %d runs in %.0f s, %d stopped by a sanitizer or with the wrong picture count" % (runs, time.time() - t0, bad))
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
