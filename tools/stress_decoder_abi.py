"""Stress for nondeterminism: decodes the same stream repeatedly through the one-picture ABI (and the batch interface)
of the given library and reports every frame whose md5 differs from the first pass.
    python tools/stress_decoder_abi.py [lib.so] [reps] [width height frames]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import build, cabi, synth  # noqa: E402

lib = sys.argv[1] if len(sys.argv) > 1 else build.LIB_HIP
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
w, h, n = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (17, 33, 14)
es, offs = synth.generate_config("cfg1_720p", n_frames=n, width=w, height=h)
oracle = cabi.decode_stream(build.LIB_ORACLE, es)[0]
bad = 0
for r in range(reps):
    for mode, size, po in ((cabi.MODE_EXPAND, None, None), (cabi.MODE_EVICT, 64 * 1024, offs)):
        try:
            got = cabi.decode_stream(lib, es, po, buffer_size=size, mode=mode)[0]
        except Exception as e:
            print("rep", r, "mode", mode, "exception", e)
            bad += 1
            continue
        if got != oracle:
            bad += 1
            diff = [i for i, (a, b) in enumerate(zip(got, oracle)) if a != b]
            print("rep %d mode %d: %d frames, differing %s" % (r, mode, len(got), diff[:8]))
print("%s: %d x 2 passes of %dx%d x %d frames, %d bad" % (os.path.basename(lib), reps, w, h, n, bad))
