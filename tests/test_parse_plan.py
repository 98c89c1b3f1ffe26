"""What jm_launch_parse decides for a pass -- the kernel (ring service in one piece / in two halves), slices per wavefront,
the head of long slices, batches, workgroups, tickets, the header step's queue threshold -- as host arithmetic, read through
`jsmpeg_hip_debug_parse_plan` (kernels.hip jm_plan_parse: no HIP call in it, so this runs without a GPU).  The figures are the
benchmark shapes' (slices, the engine's count of long slices, compressed bytes per macroblock x 16 as `JSMPEG_HIP_PARSE_SAY`
prints them on the box); the rules are the measured ones of profiles/r05_parse_notes.md sections 7 and 10."""
import ctypes

import pytest

from jsmpeg_amd import build

KEYS = ("split", "lanes", "batches", "groups", "tickets", "t_cold", "head_lanes", "head_batches", "head_end", "waves")


@pytest.fixture(scope="module")
def plan():
    lib = build.load_hip_library()
    f = lib.jsmpeg_hip_debug_parse_plan
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_uint32] * 3 + [ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]

    def run(n_slices, long_slices, bytes_per_mb_x16, with_tickets=1):
        out = (ctypes.c_uint32 * 12)()
        assert f(n_slices, long_slices, bytes_per_mb_x16, with_tickets, out) == 0
        return dict(zip(KEYS, list(out)))
    return run


def covers_every_slice_once(p, n_slices):
    """batches x their slices reach every slice: the head's batches take head_lanes each up to head_end, the rest `lanes`"""
    tail = -(-(n_slices - p["head_end"]) // p["lanes"])
    assert p["head_batches"] + tail == p["batches"]
    assert p["head_end"] <= p["head_batches"] * p["head_lanes"] and (p["head_batches"] > 0 or p["head_end"] == 0)
    if not p["tickets"]:
        assert p["groups"] * p["waves"] >= p["batches"]          # every batch has a wavefront slot (batch = slot x groups + group)


def test_the_headline_pass_draws_tickets_and_keeps_the_one_piece_service(plan):
    p = plan(522240, 48960, 126)                                  # cfg2: 64 streams x 120 pictures of 1080p, 7.9 bytes per macroblock
    assert (p["split"], p["lanes"], p["batches"], p["groups"], p["tickets"], p["t_cold"], p["head_batches"]) == (0, 64, 8160, 512, 1, 24, 0)
    covers_every_slice_once(p, 522240)


def test_dense_passes_take_the_two_halves_service_and_the_low_header_threshold(plan):
    for n, longs, bpm in ((207360, 19440, 277), (288000, 0, 336), (51840, 4860, 278)):      # 2160p 64 x 24, 320x240 intra 64 x 300, 2160p 16 x 24
        p = plan(n, longs, bpm)
        assert p["split"] == 1 and p["t_cold"] == 14, p
        covers_every_slice_once(p, n)
    assert plan(345600, 32400, 126)["split"] == 0                 # 720p 64 x 120: sparse like cfg2
    assert plan(207360, 19440, 12 * 16)["split"] == 1 and plan(207360, 19440, 12 * 16 - 1)["split"] == 0     # the cut: 12 bytes per macroblock


def test_a_pass_between_one_and_two_workgroups_per_cu_is_launched_as_two(plan):
    p = plan(207360, 19440, 277)                                  # 3240 batches: 405 workgroups of 8 would leave 149 CUs with two and 107 with one
    assert (p["batches"], p["groups"], p["tickets"]) == (3240, 512, 0)
    p = plan(103680, 9720, 277)                                   # 2160p 32 x 24: the long slices 8 per wavefront -> 2684 batches
    assert p["head_lanes"] == 8 and p["batches"] > 2048 and (p["groups"], p["tickets"]) == (512, 0)
    covers_every_slice_once(p, 103680)
    p = plan(51840, 4860, 278)                                    # 2160p 16 x 24: fewer than one workgroup per CU stays packed
    assert (p["head_lanes"], p["batches"], p["groups"]) == (4, 1950, 244)


def test_the_smallest_passes_take_a_workgroup_per_batch(plan):
    p = plan(68, 0, 0, with_tickets=0)                            # one 1080p picture through the one-picture interface
    assert (p["lanes"], p["batches"], p["groups"], p["tickets"], p["split"]) == (1, 68, 68, 0, 0)
    p = plan(16200, 1530, 126)                                    # one 720p stream of 360 pictures: 32 per wavefront behind a head of single slices
    assert (p["lanes"], p["head_lanes"], p["head_batches"], p["groups"]) == (32, 1, 1530, 249)
    covers_every_slice_once(p, 16200)


def test_without_a_ticket_counter_a_large_pass_is_one_round_of_workgroups(plan):
    p = plan(522240, 48960, 126, with_tickets=0)
    assert p["tickets"] == 0 and p["groups"] == 1020 and p["groups"] * p["waves"] >= p["batches"]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4095, 4096, 32768, 32769, 131072, 131073, 262144, 262145, 1 << 20])
def test_every_size_is_covered(plan, n):
    for longs in (0, n // 10):
        for bpm in (0, 126, 400):
            p = plan(n, longs, bpm)
            covers_every_slice_once(p, n)
            assert 1 <= p["lanes"] <= 64 and p["groups"] >= 1 and (p["tickets"] == 0 or p["groups"] == 512)
