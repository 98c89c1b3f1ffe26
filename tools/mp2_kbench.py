"""Kernel-timing experiments on the MP2 stage (no parity gate): decode the benchmark batch's audio a few times and print
the engine's phase timings.  For -D experiment builds whose output is deliberately wrong (tools/variants.sh).
    python tools/mp2_kbench.py [streams] [frames] [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import mp2, synth  # noqa: E402

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 154
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 9
streams = [synth.generate_mp2_config("mp2_stereo_44k_192", frames, stream=s)[0] for s in range(n_streams)]
with mp2.Mp2Batch(n_streams, sum(len(s) for s in streams) + 64) as b:
    b.upload(streams)
    b.decode()
    t = []
    for _ in range(reps):
        b.decode()
        t.append(b.timings())
    print({k: round(float(np.median([x[k] for x in t])), 4) for k in t[0]})
