"""The MP2 kernels' LIVE placement (C ABI part 6, jsmpeg_amd/csrc/mp2_live.hip: streams that go on from tick to tick -- a
stream's matrixing vectors in its own ring at their absolute sub-block numbers, `cap` frame places per stream and launch)
through the TEST-ONLY simulator (tests/sim/sim_mp2.cpp: the same mp2_wg_* bodies compiled by g++), against the golden
fixtures and the oracle, in the build container.  The host runtime around it is covered by the `-m gpu` tests."""
import numpy as np
import pytest

from jsmpeg_amd import cabi, synth
from mp2_util import FIXTURES, FIXTURE_IDS, SimLive, frame_md5, load_case, same_bits


@pytest.mark.parametrize("name", ["stereo_44k_192", "varying_44k", "mono_32k_48"])
@pytest.mark.parametrize("cap", [1, 3])
def test_frame_by_frame_matches_golden(name, cap):
    """One frame written per tick (what ts.js + the Player's loop give at one PES per frame): the synthesis state crosses
    every tick boundary in the stream's ring."""
    fx, data, offs = load_case(FIXTURES[FIXTURE_IDS.index(name)])
    live = SimLive(1, cap)
    got = []
    bounds = list(offs) + [len(data)]
    for k in range(fx["n_frames"]):
        live.write(0, data[int(bounds[k]):int(bounds[k + 1])])
        (pcm,) = live.tick()
        assert len(pcm) == 1
        got.append(pcm[0])
    assert frame_md5(got) == fx["frame_md5"]


def test_ragged_pieces_of_several_streams(libs):
    """Streams of every generator configuration fed in random pieces (frames cut anywhere, ticks that find nothing complete,
    ticks that find more frames than a launch has places for): per stream the frames of the whole stream decoded in one piece."""
    rng = np.random.RandomState(20260930)
    streams = [synth.generate_mp2_config(name, 9 + 4 * i, stream=70 + i)[0] for i, name in enumerate(synth.MP2_CONFIGS)]
    want = [cabi.decode_mp2_stream(libs["oracle"], s)[0] for s in streams]
    n = len(streams)
    live = SimLive(n, 2)
    at = [0] * n
    got = [[] for _ in range(n)]
    for _ in range(400):
        for s in range(n):
            if at[s] < len(streams[s]) and rng.randint(3):
                k = int(rng.choice([1, 7, 100, 417, 1500, 4000]))
                live.write(s, streams[s][at[s]:at[s] + k])
                at[s] += k
        for s, pcm in enumerate(live.tick()):
            got[s].extend(pcm)
        if all(at[s] >= len(streams[s]) and not live.store[s] for s in range(n)):
            break
    for s in range(n):
        assert len(got[s]) == len(want[s]) and same_bits(np.array(got[s]), want[s]), s
        assert not live.store[s]


def test_a_stream_that_stops_at_a_header_the_reference_refuses(libs):
    """Noise behind three frames: the walk stops there in every tick, the frames in front are decoded, nothing after."""
    data, offs = synth.generate_mp2_config("mp2_stereo_44k_192", 6, stream=5)
    bad = np.concatenate([data[:int(offs[3])], np.frombuffer(b"\x12\x34\x56\x78" * 50, np.uint8), data[int(offs[3]):]])
    want = cabi.decode_mp2_stream(libs["oracle"], data)[0]
    live = SimLive(1, 8)
    live.write(0, bad)
    (pcm,) = live.tick()
    assert same_bits(pcm, want[:3])
    (pcm,) = live.tick()
    assert len(pcm) == 0 and len(live.store[0]) == len(bad) - int(offs[3])


def test_the_sub_block_count_near_its_ceiling():
    """A stream that has been playing for 200 hours: the count of sub-blocks synthesised is kept below 2^31 by steps of 2^29 (a
    multiple of 16 and of every ring size: the same ring slots, the same phase of the reference's v_pos) -- the samples across
    the step are the golden ones."""
    fx, data, offs = load_case(FIXTURES[FIXTURE_IDS.index("varying_44k")])
    start = (1 << 30) + (1 << 29) - 36 * 4 - 16          # (a multiple of 16; the step falls behind the stream's fourth or fifth frame)
    assert start % 16 == 0
    live = SimLive(1, 2, n_abs0=start)
    bounds = list(offs) + [len(data)]
    got, stepped = [], False
    for k in range(fx["n_frames"]):
        before = int(live.n_abs[0])
        live.write(0, data[int(bounds[k]):int(bounds[k + 1])])
        (pcm,) = live.tick()
        stepped = stepped or int(live.n_abs[0]) < before
        got.append(pcm[0])
    assert stepped and frame_md5(got) == fx["frame_md5"]
