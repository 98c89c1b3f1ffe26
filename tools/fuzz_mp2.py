"""Randomised parity sweep of the MP2 stage on the GPU: random Layer II generator configurations (every sampling
frequency, bit rates / modes / CRC / padding changing from frame to frame, forbidden-but-decodable codes, sparse and
dense allocations), intact and damaged, in batches through jsmpeg_hip_mp2_batch_* and a sample through the one-frame
ABI, PCM against the oracle (checker) bit for bit.     python tools/fuzz_mp2.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from jsmpeg_amd import build, cabi, mp2, synth  # noqa: E402
from test_mp2_sim_device_functions import _damaged  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
t0 = time.time()
bad = frames = 0
for first in range(0, cases, 200):
    streams, want, full = [], [], []
    for case in range(first, min(first + 200, cases)):
        kw = dict(sample_rate_index=int(rng.randint(0, 3)), bitrate_index=int(rng.randint(1, 15)), mode=int(rng.randint(0, 4)),
                  crc=int(rng.randint(0, 2)), vary=int(rng.rand() < 0.5), quirks=int(rng.rand() < 0.3),
                  alloc_permille=int(rng.choice([150, 500, 800, 1000])), sf_lo=int(rng.choice([8, 12, 30])), sf_hi=62,
                  seed=int(rng.randint(1, 2 ** 31 - 1)))
        data = synth.generate_mp2(int(rng.randint(1, 14)), **kw)[0]
        if case % 2:
            data = _damaged(rng, data)
        pcm, idx, sizes, _ = cabi.decode_mp2_stream(oracle, data)
        full.append((pcm, idx))
        if len(pcm) and sum(sizes) > len(data):
            pcm = pcm[:-1]                       # a last frame that is not all there: the batch does not decode it
        streams.append(data)
        want.append(pcm)
    with mp2.Mp2Batch(len(streams), sum(len(s) for s in streams) + 64) as b:
        b.upload(streams)
        n = b.decode()
        frames += n
        if n != sum(len(w) for w in want):
            bad += 1
            print("batch from case %d: %d frames, expected %d" % (first, n, sum(len(w) for w in want)))
        for i, w in enumerate(want):
            got = b.read_pcm(i) if b.frame_count(i) else np.zeros((0, 2, 1152), np.float32)
            if got.shape != w.shape or not np.array_equal(got.view(np.uint32), w.view(np.uint32)):
                bad += 1
                print("case %d: batch PCM differs" % (first + i))
    for i in range(0, len(streams), 16):
        pcm, idx, _, _ = cabi.decode_mp2_stream(build.LIB_HIP, streams[i])
        if idx != full[i][1] or pcm.shape != full[i][0].shape or not np.array_equal(pcm.view(np.uint32), full[i][0].view(np.uint32)):
            bad += 1
            print("case %d: one-frame ABI differs" % (first + i))
print("%d cases (half of them damaged), %d frames, %d mismatches in %.1fs" % (cases, frames, bad, time.time() - t0))
sys.exit(1 if bad else 0)
