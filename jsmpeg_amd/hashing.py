"""Host mirror of the device frame hash (csrc/kernels.hip k_hash):
h = sum_i mix(word_i, i) mod 2**64 over the little-endian 64-bit words of Y|Cr|Cb."""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M = np.uint64(0xD6E8FEB86659FD93)


def frame_hash(*planes):
    data = np.concatenate([np.ascontiguousarray(p, dtype=np.uint8).ravel() for p in planes])
    assert data.size % 8 == 0
    w = data.view("<u8")
    with np.errstate(over="ignore"):
        i = np.arange(1, w.size + 1, dtype=np.uint64)
        t = w ^ (i * _G)
        t = t * _M
        t ^= t >> np.uint64(32)
        t = t * _M
        return int(np.sum(t, dtype=np.uint64))
