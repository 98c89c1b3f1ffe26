"""ctypes front-end of the synthetic MPEG-1 ES generator / TS muxer
(csrc/synth_es.c).  Configs follow SURVEY.md section 8d."""
import ctypes
import os

import numpy as np

from . import build as _build


class SynthParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("width", "height", "n_frames", "gop")] + \
               [("seed", ctypes.c_uint32)] + \
               [(n, ctypes.c_int32) for n in ("ac_max", "qscale_lo", "qscale_hi", "escape_permille",
                                               "custom_quant", "quirk_levels", "dc_size_max",
                                               "coded_permille", "f_code_max", "syntax_quirks", "mv_jitter")]


class SynthStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("macroblocks", "predicted", "coded_blocks", "coefficients")]


class SynthMp2Params(ctypes.Structure):
    _fields_ = [("n_frames", ctypes.c_int32), ("seed", ctypes.c_uint32)] + \
               [(n, ctypes.c_int32) for n in ("sample_rate_index", "bitrate_index", "mode", "crc", "sf_lo", "sf_hi",
                                               "alloc_permille", "vary", "quirks")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.LIB_SYNTH
        if not os.path.exists(path):
            _build.build_synth()
        _lib = ctypes.CDLL(path)
        _lib.synth_es_generate.restype = ctypes.c_size_t
        _lib.synth_es_generate.argtypes = [ctypes.POINTER(SynthParams), ctypes.c_void_p, ctypes.c_size_t,
                                           ctypes.c_void_p, ctypes.POINTER(SynthStats)]
        _lib.synth_ts_mux.restype = ctypes.c_size_t
        _lib.synth_ts_mux.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
                                      ctypes.c_void_p, ctypes.c_size_t]
        _lib.synth_mp2_generate.restype = ctypes.c_size_t
        _lib.synth_mp2_generate.argtypes = [ctypes.POINTER(SynthMp2Params), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    return _lib


BASE_SEED = 0x4A534D50  # 'JSMP'

# name -> generator parameters.  `streams`/`frames` are the
# full-size figures; tests pass smaller n_frames.
CONFIGS = {
    "cfg0_240p_intra": dict(width=320, height=240, gop=1, frames=300, ac_max=6, qscale_lo=4, qscale_hi=11,
                            escape_permille=20, dc_size_max=3, coded_permille=500, f_code_max=1, cfg=0),
    "cfg1_720p": dict(width=1280, height=720, gop=12, frames=360, ac_max=3, qscale_lo=4, qscale_hi=11,
                      escape_permille=20, dc_size_max=3, coded_permille=400, f_code_max=3, cfg=1),
    "cfg2_1080p": dict(width=1920, height=1080, gop=12, frames=120, ac_max=3, qscale_lo=4, qscale_hi=11,
                       escape_permille=20, dc_size_max=3, coded_permille=400, f_code_max=3, cfg=2),
    "cfg4_2160p": dict(width=3840, height=2160, gop=12, frames=24, ac_max=8, qscale_lo=2, qscale_hi=6,
                       escape_permille=20, dc_size_max=5, coded_permille=600, f_code_max=3, cfg=4),
}


def generate_es(width, height, n_frames, gop=12, seed=BASE_SEED, ac_max=4, qscale_lo=4, qscale_hi=11,
                escape_permille=20, custom_quant=0, quirk_levels=0, dc_size_max=3, coded_permille=400,
                f_code_max=3, syntax_quirks=0, mv_jitter=0, with_stats=False):
    """Returns (es_bytes: np.uint8[n], pic_offsets: np.uint32[n_frames+1]) [+ stats dict]."""
    p = SynthParams(width, height, n_frames, gop, seed & 0xFFFFFFFF, ac_max, qscale_lo, qscale_hi,
                    escape_permille, custom_quant, quirk_levels, dc_size_max, coded_permille, f_code_max, syntax_quirks, mv_jitter)
    mbs = ((width + 15) // 16) * ((height + 15) // 16)
    cap = 4096 + n_frames * (mbs * (64 + 40 * max(ac_max, 1)) + 4096)
    buf = np.empty(cap, dtype=np.uint8)
    offs = np.zeros(n_frames + 1, dtype=np.uint32)
    st = SynthStats()
    n = lib().synth_es_generate(ctypes.byref(p), buf.ctypes.data, cap, offs.ctypes.data, ctypes.byref(st))
    if n == 0:
        raise RuntimeError("synthetic ES generation overflowed its buffer")
    if with_stats:
        return buf[:n].copy(), offs, {k: int(getattr(st, k)) for k, _ in SynthStats._fields_}
    return buf[:n].copy(), offs


def stuff_zero_bytes(es, pic_offsets, before_pictures=0, before_slices=0):
    """zero_byte stuffing in front of start codes (valid MPEG-1: next_start_code() skips any number of zero bytes; what
    a CBR encoder pads pictures with): `before_pictures` zero bytes in front of every picture start code but the
    stream's first, `before_slices` in front of every slice start code but a picture's first.  The counts cycle 1..n
    from code to code, so a stream holds every amount up to n.  Returns (es, pic_offsets) with the offsets moved so
    that a picture's range still begins with its start code (the stuffing belongs to the range before it)."""
    es = np.ascontiguousarray(es, dtype=np.uint8)
    z = (es[:-3] == 0) & (es[1:-2] == 0) & (es[2:-1] == 1)
    pos = np.flatnonzero(z)
    code = es[pos + 3]
    ins = np.zeros(len(pos), dtype=np.int64)
    k_pic = k_sl = 0
    seen_picture = False
    prev_code = -1
    for i, (p, c) in enumerate(zip(pos, code)):
        if c == 0x00:
            if seen_picture and before_pictures:
                ins[i] = 1 + k_pic % before_pictures
                k_pic += 1
            seen_picture = True
        elif 0x01 <= c <= 0xAF and before_slices and 0x01 <= prev_code <= 0xAF:
            ins[i] = 1 + k_sl % before_slices
            k_sl += 1
        prev_code = int(c)
    out = np.zeros(len(es) + int(ins.sum()), dtype=np.uint8)
    shift = np.zeros(len(es) + 1, dtype=np.int64)
    np.add.at(shift, pos, ins)
    shift = np.cumsum(shift)
    out[np.arange(len(es)) + shift[:len(es)]] = es
    offs = np.asarray(pic_offsets, dtype=np.int64)
    new_offs = offs + shift[np.minimum(offs, len(es))]
    return out, new_offs.astype(np.uint32)


ENCODED_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def generate_config(name, n_frames=None, stream=0, stuff_pictures=0, stuff_slices=0, **overrides):
    if name.startswith("enc:"):
        # not generated here: a stream written by the independent test-side encoder (tests/enc/mpeg1_enc.py), committed as
        # tests/golden/<name>.m1v + .offsets.npy so that fixtures made from it mean the same bytes everywhere
        es = np.fromfile(os.path.join(ENCODED_DIR, name[4:] + ".m1v"), dtype=np.uint8)
        offs = np.load(os.path.join(ENCODED_DIR, name[4:] + ".offsets.npy")).astype(np.uint32)
        assert n_frames in (None, len(offs) - 1) and not overrides, "an encoded stream is what it is"
        return es, offs
    c = dict(CONFIGS[name])
    cfg = c.pop("cfg")
    frames = c.pop("frames")
    c.update(overrides)
    seed = (BASE_SEED + cfg + 7919 * stream) & 0xFFFFFFFF
    out = generate_es(n_frames=n_frames or frames, seed=seed, **c)
    if stuff_pictures or stuff_slices:
        es, offs = stuff_zero_bytes(out[0], out[1], stuff_pictures, stuff_slices)
        out = (es, offs) + tuple(out[2:])
    return out


def mux_ts(es, pic_offsets, fps=30.0):
    n_pics = len(pic_offsets) - 1
    cap = (len(es) // 170 + 2 * n_pics + 16) * 188
    out = np.empty(cap, dtype=np.uint8)
    es = np.ascontiguousarray(es)
    offs = np.ascontiguousarray(pic_offsets, dtype=np.uint32)
    n = lib().synth_ts_mux(es.ctypes.data, offs.ctypes.data, n_pics, float(fps), out.ctypes.data, cap)
    if n == 0:
        raise RuntimeError("TS mux overflowed its buffer")
    return out[:n].copy()


# MP2 audio.  name -> generator parameters; bitrate_index is the header value (1..14:
# 32 48 56 64 80 96 112 128 160 192 224 256 320 384 kbit/s), sample_rate_index 0 / 1 / 2 = 44.1 / 48 / 32 kHz.
MP2_CONFIGS = {
    "mp2_stereo_44k_192": dict(sample_rate_index=0, bitrate_index=10, mode=0, crc=0),
    "mp2_joint_48k_128": dict(sample_rate_index=1, bitrate_index=8, mode=1, crc=1),
    "mp2_mono_32k_48": dict(sample_rate_index=2, bitrate_index=2, mode=3, crc=0),
    "mp2_dual_44k_384": dict(sample_rate_index=0, bitrate_index=14, mode=2, crc=1, alloc_permille=950),
    "mp2_mono_48k_64": dict(sample_rate_index=1, bitrate_index=4, mode=3, crc=1),
    "mp2_varying_44k": dict(sample_rate_index=0, bitrate_index=8, mode=0, crc=0, vary=1),
    "mp2_varying_32k_quirks": dict(sample_rate_index=2, bitrate_index=8, mode=0, crc=0, vary=1, quirks=1),
}


def generate_mp2(n_frames, sample_rate_index=0, bitrate_index=10, mode=0, crc=0, sf_lo=9, sf_hi=62,
                 alloc_permille=800, vary=0, quirks=0, seed=BASE_SEED + 0x4D5032):
    """Returns (mp2 bytes: np.uint8[n], frame_offsets: np.uint32[n_frames + 1])."""
    p = SynthMp2Params(n_frames, seed & 0xFFFFFFFF, sample_rate_index, bitrate_index, mode, crc, sf_lo, sf_hi,
                       alloc_permille, vary, quirks)
    cap = n_frames * 1800 + 64
    buf = np.empty(cap, dtype=np.uint8)
    offs = np.zeros(n_frames + 1, dtype=np.uint32)
    n = lib().synth_mp2_generate(ctypes.byref(p), buf.ctypes.data, cap, offs.ctypes.data)
    if n == 0:
        raise RuntimeError("synthetic MP2 generation failed")
    return buf[:n].copy(), offs


def generate_mp2_config(name, n_frames, stream=0, **overrides):
    c = dict(MP2_CONFIGS[name])
    c.update(overrides)
    c.setdefault("seed", (BASE_SEED + 0x4D5032 + 7919 * stream) & 0xFFFFFFFF)
    return generate_mp2(n_frames, **c)
