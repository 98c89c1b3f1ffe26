"""The device start-code scan (k_scan: one pass, chained scan with look-back) and the slice order (k_order_*) against
numpy on crafted byte soups -- dense runs of codes, codes across every lane / piece / chunk boundary of the kernel,
zero stuffing, truncated codes at the end.  Reference: the serial byte loop of src/wasm/buffer.c:73-110."""
import ctypes

import numpy as np
import pytest

from jsmpeg_amd import batch as jb
from jsmpeg_amd.distributed import find_start_codes

pytestmark = pytest.mark.gpu

BASE = 16          # a batch's first stream starts 16 bytes into the ES buffer


def soup(n, seed, plant):
    rng = np.random.default_rng(seed)
    # enough zeros and ones that 00 00, 00 00 01 and whole start codes (picture codes too) turn up by themselves
    es = rng.choice(np.array([0, 1, 2, 3, 9, 0x47, 0x55, 0xB3, 0xff, 0x80, 0xAF, 0xB0], np.uint8), size=n).astype(np.uint8)
    codes = np.array([0x01, 0x02, 0x44, 0xAF, 0xB0, 0xB5, 0xB8, 0xB2, 0xE0], np.uint8)      # no picture / sequence codes
    for p in plant:
        if 0 <= p and p + 4 <= n:
            es[p:p + 4] = (0, 0, 1, codes[p % len(codes)])
    return es


def boundaries(n):
    """positions (in the stream) around every boundary the kernel has: 64-byte lane pieces, 16 KiB pieces, chunks"""
    out = []
    for unit in (64, 16384, 7 * 16384, 2 * 16384, 3 * 16384):
        for k in range(1, min(40, n // unit + 1)):
            edge = k * unit - BASE
            out += [edge - 4, edge - 3, edge - 2, edge - 1, edge, edge + 1]
    return out


def device_tables(es, max_pictures=20000, size=(64, 48)):
    L = jb.lib()
    L.jsmpeg_hip_batch_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
    with jb.Batch(size[0], size[1], 1, max_pictures, len(es) + 4096) as b:
        b.upload([es])
        b.decode()
        c = b.counters()

        def rd(what, dtype, count):
            a = np.zeros(max(count, 1), dtype)
            assert L.jsmpeg_hip_batch_debug_read(b.h, what, a.ctypes.data, 0, a.nbytes) == 0, jb.last_error()
            return a[:count]
        n, ns = c["start_codes"], c["slice_codes"]
        return (c, rd(0, np.uint32, n).astype(np.int64) - BASE, rd(1, np.uint8, n), rd(9, np.uint32, ns), rd(10, np.uint32, ns),
                rd(2, np.uint32, n))


@pytest.mark.parametrize("n,seed", [(1, 1), (3, 2), (4, 3), (63, 4), (64, 5), (5000, 6), (16384 - BASE, 7), (16384 - BASE + 3, 8),
                                    (300_000, 9), (3_000_000, 10)])
def test_scan_matches_numpy(n, seed):
    plant = boundaries(n) + list(range(1000, 1256, 4)) + [n - 4, n - 3, n - 5, 0, 1, 7]
    es = soup(n, seed, plant)
    pos, code = find_start_codes(es)
    c, g_pos, g_code, g_slices, g_order, g_owner = device_tables(es)
    assert c["start_codes"] == len(pos)
    assert np.array_equal(g_pos, pos) and np.array_equal(g_code, code)
    is_slice = (code >= 1) & (code <= 0xAF)
    assert np.array_equal(g_slices, np.nonzero(is_slice)[0])              # the slice list: stream order
    assert np.array_equal(np.sort(g_order), np.sort(g_slices))            # the parse order: a permutation of it
    assert c["pictures"] == int((code == 0).sum())


def test_scan_many_pieces_per_chunk():
    """an input large enough for several pieces per ticket (32 MiB per piece of the chunk size)"""
    n = 70_000_000
    rng = np.random.default_rng(11)
    plant = rng.integers(0, n - 4, size=200_000).tolist() + boundaries(n)
    es = soup(n, 12, plant)
    pos, code = find_start_codes(es)
    c, g_pos, g_code, g_slices, g_order, _ = device_tables(es)
    assert c["start_codes"] == len(pos)
    assert np.array_equal(g_pos, pos) and np.array_equal(g_code, code)
    assert np.array_equal(g_slices, np.nonzero((code >= 1) & (code <= 0xAF))[0])
    assert np.array_equal(np.sort(g_order), np.sort(g_slices))


def test_slice_order_is_longest_first():
    """real slices: the order the parse takes them in is by length (1024 bins), longest first"""
    from jsmpeg_amd import synth
    es, _ = synth.generate_config("cfg1_720p", n_frames=24)
    pos, code = find_start_codes(es)
    c, g_pos, g_code, g_slices, g_order, g_owner = device_tables(es, max_pictures=32, size=(1280, 720))
    assert c["decoded"] == 24 and c["slices"] == c["slice_codes"] == len(g_order)
    end = np.append(pos[1:], len(es))
    length = (end - pos)[g_order]
    mean = len(es) // len(g_order)
    shift = 0
    while (mean >> shift) >= 512:
        shift += 1
    bins = np.minimum(1 + (length >> shift), 1023)
    assert (np.diff(bins) <= 0).all()
    assert length[:40].mean() > length[-40:].mean()


@pytest.mark.parametrize("code", [0xB5, 0x01, 0x00])
def test_more_start_codes_than_the_tables_hold(code):
    """a start code every four bytes (more than one per 16 bytes of capacity): the pass ends with an error, not a fault,
    and the batch decodes the next input"""
    from jsmpeg_amd import synth
    n = 1 << 20
    es = np.tile(np.array([0, 0, 1, code], np.uint8), n // 4)
    good, _ = synth.generate_config("cfg1_720p", n_frames=3)
    with jb.Batch(1280, 720, 1, 64, n + 4096) as b:
        b.upload([es])
        with pytest.raises(RuntimeError, match="overflow"):
            b.decode()
        b.upload([good])
        assert b.decode() == 3 and b.counters()["decoded"] == 3
