"""The workgroup bodies of the MP2 kernels (jsmpeg_amd/csrc/mp2_dev.h: mp2_wg_*), compiled by g++ into a TEST-ONLY
simulator (tests/sim/sim_mp2.cpp), against the golden fixtures and the oracle -- so their logic AND their
arithmetic (same source, -ffp-contract=off) are checked in the build container, which has no GPU.  The product
never runs them on the CPU; launch shapes, LDS staging and the host runtime are covered by the `-m gpu` tests."""
import ctypes

import numpy as np
import pytest

from jsmpeg_amd import cabi, synth
from mp2_util import FIXTURES, FIXTURE_IDS, frame_md5, load_case, same_bits, sim_batch, sim_lib


@pytest.mark.parametrize("path", FIXTURES, ids=FIXTURE_IDS)
def test_batch_pipeline_matches_golden(path):
    fx, data, offs = load_case(path)
    (pcm,) = sim_batch([data])
    assert frame_md5(pcm) == fx["frame_md5"]


def test_batch_of_unequal_streams(libs):
    """Several streams of different lengths and configurations in one batch: lookback never crosses a stream
    boundary, every stream starts from a silent synthesis state."""
    streams = [synth.generate_mp2_config(name, 5 + 7 * i, stream=40 + i)[0] for i, name in enumerate(synth.MP2_CONFIGS)]
    streams.insert(2, np.zeros(0, np.uint8))                       # an empty stream
    streams.append(streams[0][:len(streams[0]) - 100])             # last frame cut: not decoded
    streams.append(np.frombuffer(b"\x00" * 300, dtype=np.uint8))   # no frame at all
    got = sim_batch(streams)
    for s, g in zip(streams, got):
        want, _, sizes, _ = cabi.decode_mp2_stream(libs["oracle"], s)
        if len(s) and len(want) and sum(sizes) > len(s):           # the oracle decodes a cut last frame (zeros), the batch does not
            want = want[:-1]
        assert same_bits(g, want)
    assert len(got[2]) == 0 and len(got[-1]) == 0 and len(got[-2]) == len(got[0]) - 1


@pytest.mark.parametrize("name", ["varying_44k", "dual_44k_384", "mono_32k_48"])
def test_ring_mode_frame_by_frame(name, libs):
    """The decoder ABI's path: one frame per launch through the 64-vector ring, state carried in the ring and the
    sub-block count only."""
    fx, data, offs = load_case(FIXTURES[FIXTURE_IDS.index(name)])
    lib = sim_lib()
    ring = np.zeros(64 * 64, np.float32)
    n_abs = ctypes.c_uint32(0)
    out = np.zeros((fx["n_frames"], 2, 1152), np.float32)
    for k in range(fx["n_frames"]):
        frame = np.ascontiguousarray(data[int(offs[k]):int(offs[k]) + 4592])      # what the frame's fields can reach, like the ABI stages it
        lib.sim_mp2_ring_frame(ctypes.c_void_p(frame.ctypes.data), len(frame), ctypes.c_void_p(ring.ctypes.data),
                               ctypes.byref(n_abs), ctypes.c_void_p(out[k].ctypes.data))
    assert n_abs.value == 36 * fx["n_frames"]
    assert frame_md5(out) == fx["frame_md5"]


def test_randomised_sweep_against_the_oracle(libs):
    """150 random generator configurations (every sampling frequency, bit rates / modes / CRC / padding changing from
    frame to frame, forbidden-but-decodable codes, sparse and dense allocations, quiet and loud scalefactors) through
    the device functions in one batch each; every sample against the oracle."""
    rng = np.random.RandomState(20260923)
    checked = 0
    for case in range(150):
        kw = dict(sample_rate_index=int(rng.randint(0, 3)), bitrate_index=int(rng.randint(1, 15)), mode=int(rng.randint(0, 4)),
                  crc=int(rng.randint(0, 2)), vary=int(rng.rand() < 0.5), quirks=int(rng.rand() < 0.3),
                  alloc_permille=int(rng.choice([150, 500, 800, 1000])), sf_lo=int(rng.choice([8, 12, 30])), sf_hi=62,
                  seed=int(rng.randint(1, 2 ** 31 - 1)))
        n_frames = int(rng.randint(1, 9))
        data, _ = synth.generate_mp2(n_frames, **kw)
        want = cabi.decode_mp2_stream(libs["oracle"], data)[0]
        (got,) = sim_batch([data])
        assert len(want) == n_frames and same_bits(got, want), (case, kw)
        checked += n_frames
    assert checked > 400


def _damaged(rng, data):
    bad = data.copy()
    kind = rng.randint(0, 4)
    if kind == 0:                                   # scattered bit flips
        for _ in range(int(rng.randint(1, 30))):
            bad[int(rng.randint(0, len(bad)))] ^= 1 << int(rng.randint(0, 8))
    elif kind == 1:                                 # a run of random bytes
        a = int(rng.randint(0, len(bad) - 8))
        n = int(rng.randint(1, min(400, len(bad) - a)))
        bad[a:a + n] = rng.randint(0, 256, n).astype(np.uint8)
    elif kind == 2:                                 # truncated anywhere
        bad = bad[:int(rng.randint(1, len(bad)))]
    else:                                           # bytes removed from the middle (every later header moves)
        a = int(rng.randint(4, len(bad) - 4))
        bad = np.concatenate([bad[:a], bad[a + int(rng.randint(1, 4)):]])
    return np.ascontiguousarray(bad)


def test_damaged_streams_still_equal_the_oracle(libs):
    """Bit flips, random runs, truncation, dropped bytes: the batch pipeline decodes exactly the frames the
    reference's decode() loop would reach and stops where it stops -- damaged allocation / scalefactor / sample
    fields change the samples, never the agreement with the oracle."""
    rng = np.random.RandomState(99)
    streams, want = [], []
    for case in range(120):
        data, _ = synth.generate_mp2_config(list(synth.MP2_CONFIGS)[case % len(synth.MP2_CONFIGS)], 6, stream=500 + case)
        bad = _damaged(rng, data)
        pcm, _, sizes, _ = cabi.decode_mp2_stream(libs["oracle"], bad)
        if len(pcm) and sum(sizes) > len(bad):
            pcm = pcm[:-1]                          # a last frame that is not all there: not decoded by the batch
        streams.append(bad)
        want.append(pcm)
    got = sim_batch(streams)
    assert sum(len(w) for w in want) > 200
    for i in range(len(streams)):
        assert same_bits(got[i], want[i]), i


def _overcommitted_stream(seed):
    """A frame whose allocation promises far more sample bits than its length holds (every subband 16-bit samples at
    112 kbit/s): the reference reads on into the bytes that follow; 6 KB of random bytes follow."""
    rng = np.random.RandomState(seed)
    bits = [1] * 11 + [1, 1] + [1, 0] + [1]            # sync, MPEG-1, Layer II, no CRC
    bits += [0, 1, 1, 1] + [0, 0] + [0] + [0] + [0, 0] + [0, 0] + [0, 0, 0, 0]   # 112 kbit/s, 44.1 kHz, stereo
    bits += [1] * (2 * (11 * 4 + 12 * 3 + 7 * 2))      # allocation: every code all ones -> 65535 steps everywhere
    body = np.packbits(np.array(bits, np.uint8))
    return np.concatenate([body, rng.randint(0, 256, 6000).astype(np.uint8)])


def test_frame_that_promises_more_bits_than_it_has(libs):
    for seed in range(6):
        data = _overcommitted_stream(seed)
        want, _, sizes, _ = cabi.decode_mp2_stream(libs["oracle"], data, max_frames=1)
        assert len(want) == 1 and sizes[0] == 365
        (got,) = sim_batch([data])
        assert len(got) >= 1 and same_bits(got[:1], want), seed
