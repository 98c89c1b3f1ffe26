"""ctypes view of the 15-function decoder C ABI (reference src/wasm/mpeg1.h:10-25).

The same wrapper drives three libraries that export that ABI:
  * jsmpeg_amd/libjsmpeg_hip.so  - the product (HIP, gfx950)
  * oracle/libmpeg1_oracle.so    - CPU restatement (tests only)
  * oracle/_ref/libjsmpeg_ref.so - the reference's own C (tests / cpu_baseline only)
This module never picks a library itself: callers pass the path."""
import ctypes
import os

import numpy as np

MODE_EVICT = 1
MODE_EXPAND = 2

_SIGS = {
    "mpeg1_decoder_create": (ctypes.c_void_p, [ctypes.c_uint, ctypes.c_int]),
    "mpeg1_decoder_destroy": (None, [ctypes.c_void_p]),
    "mpeg1_decoder_get_write_ptr": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_uint]),
    "mpeg1_decoder_get_index": (ctypes.c_int, [ctypes.c_void_p]),
    "mpeg1_decoder_set_index": (None, [ctypes.c_void_p, ctypes.c_uint]),
    "mpeg1_decoder_did_write": (None, [ctypes.c_void_p, ctypes.c_uint]),
    "mpeg1_decoder_has_sequence_header": (ctypes.c_int, [ctypes.c_void_p]),
    "mpeg1_decoder_get_frame_rate": (ctypes.c_float, [ctypes.c_void_p]),
    "mpeg1_decoder_get_coded_size": (ctypes.c_int, [ctypes.c_void_p]),
    "mpeg1_decoder_get_width": (ctypes.c_int, [ctypes.c_void_p]),
    "mpeg1_decoder_get_height": (ctypes.c_int, [ctypes.c_void_p]),
    "mpeg1_decoder_get_y_ptr": (ctypes.c_void_p, [ctypes.c_void_p]),
    "mpeg1_decoder_get_cr_ptr": (ctypes.c_void_p, [ctypes.c_void_p]),
    "mpeg1_decoder_get_cb_ptr": (ctypes.c_void_p, [ctypes.c_void_p]),
    "mpeg1_decoder_decode": (ctypes.c_bool, [ctypes.c_void_p]),
}
ABI_SYMBOLS = tuple(_SIGS)

_libs = {}


def _load_shared(path):
    # the HIP library goes through build.load_hip_library (one HIP runtime per process, the one PyTorch ships if installed);
    # the oracle and the reference's own C are plain CPU libraries
    from . import build
    if os.path.abspath(path) == os.path.abspath(build.LIB_HIP):
        return build.load_hip_library(path)
    return ctypes.CDLL(path)


def load(path):
    lib = _libs.get(path)
    if lib is None:
        lib = _load_shared(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _libs[path] = lib
    return lib


class Mpeg1Decoder:
    """One decoder handle of whichever library `path` names."""

    def __init__(self, path, buffer_size=512 * 1024, mode=MODE_EXPAND):
        self.lib = load(path)
        # the product library has an error channel beside the reference's 15 functions; the checker libraries do not
        self._last_error = getattr(self.lib, "jsmpeg_hip_last_error", None)
        if self._last_error is not None:
            self._last_error.restype = ctypes.c_char_p
            self._last_error.argtypes = []
        self.h = self.lib.mpeg1_decoder_create(buffer_size, mode)
        if not self.h:
            why = self._last_error() if self._last_error is not None else b""
            raise RuntimeError("mpeg1_decoder_create failed (%s)%s" % (path, ": " + why.decode("utf-8", "replace") if why else ""))

    def close(self):
        if self.h:
            self.lib.mpeg1_decoder_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def write(self, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n = int(data.size)
        ptr = self.lib.mpeg1_decoder_get_write_ptr(self.h, n)
        ctypes.memmove(ptr, data.ctypes.data, n)
        self.lib.mpeg1_decoder_did_write(self.h, n)

    def decode(self):
        """One picture.  False = no complete picture buffered.  On the product library a device failure comes back as
        false + a message in jsmpeg_hip_last_error() (include/jsmpeg_hip.h): raised here, never read as end of data."""
        got = bool(self.lib.mpeg1_decoder_decode(self.h))
        if not got and self._last_error is not None:
            msg = self._last_error()
            if msg:
                raise RuntimeError("mpeg1_decoder_decode failed: " + msg.decode("utf-8", "replace"))
        return got

    @property
    def index(self):
        return self.lib.mpeg1_decoder_get_index(self.h)

    @index.setter
    def index(self, v):
        self.lib.mpeg1_decoder_set_index(self.h, v)

    @property
    def has_sequence_header(self):
        return bool(self.lib.mpeg1_decoder_has_sequence_header(self.h))

    @property
    def frame_rate(self):
        return self.lib.mpeg1_decoder_get_frame_rate(self.h)

    @property
    def coded_size(self):
        return self.lib.mpeg1_decoder_get_coded_size(self.h)

    @property
    def width(self):
        return self.lib.mpeg1_decoder_get_width(self.h)

    @property
    def height(self):
        return self.lib.mpeg1_decoder_get_height(self.h)

    def ahead_stats(self):
        """Product only: (passes of the batch engine, pictures served from them) of the decode-ahead (include/jsmpeg_hip.h)"""
        fn = self.lib.jsmpeg_hip_decoder_ahead_stats
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        out = (ctypes.c_uint64 * 2)()
        if fn(self.h, out) != 0:
            raise RuntimeError("jsmpeg_hip_decoder_ahead_stats failed")
        return int(out[0]), int(out[1])

    def render_rgba(self):
        """Product only (libjsmpeg_hip.so): the most recently decoded picture as RGBA, converted on the
        device (jsmpeg_hip_decoder_render_rgba); uint8[height, width, 4]."""
        fn = self.lib.jsmpeg_hip_decoder_render_rgba
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        if fn(self.h, out.ctypes.data) < 0:
            raise RuntimeError("jsmpeg_hip_decoder_render_rgba failed")
        return out

    def planes(self):
        """Copies of the most recently decoded (Y, Cr, Cb) coded-size planes."""
        n = self.coded_size
        out = []
        for getter, size in ((self.lib.mpeg1_decoder_get_y_ptr, n), (self.lib.mpeg1_decoder_get_cr_ptr, n >> 2),
                             (self.lib.mpeg1_decoder_get_cb_ptr, n >> 2)):
            ptr = getter(self.h)
            out.append(np.ctypeslib.as_array((ctypes.c_uint8 * size).from_address(ptr)).copy())
        return tuple(out)


def decode_stream(path, es, pic_offsets=None, buffer_size=None, mode=MODE_EXPAND, keep="hash", max_frames=None):
    """Feeds `es` the way ts.js feeds the decoder (one write per picture when
    `pic_offsets` is given, reference src/ts.js:205-210, else one write) and
    pulls every picture.  Returns a list of per-frame md5 hex digests over
    Y|Cr|Cb (keep="hash") or of (Y, Cr, Cb) arrays (keep="planes"), plus the
    list of bit indices observed after each decode()."""
    import hashlib
    es = np.ascontiguousarray(es, dtype=np.uint8)
    frames, indices = [], []
    with Mpeg1Decoder(path, buffer_size or (len(es) + 1024), mode) as dec:
        def pull():
            while (max_frames is None or len(frames) < max_frames) and dec.decode():
                indices.append(dec.index)
                y, cr, cb = dec.planes()
                if keep == "hash":
                    h = hashlib.md5()
                    h.update(y.tobytes()); h.update(cr.tobytes()); h.update(cb.tobytes())
                    frames.append(h.hexdigest())
                else:
                    frames.append((y, cr, cb))
        if pic_offsets is None:
            dec.write(es)
            pull()
        else:
            # streaming-style: write picture k, decode what is complete
            # (a picture is only complete once the next start code is in)
            n = len(pic_offsets) - 1
            for k in range(n):
                end = len(es) if k == n - 1 else int(pic_offsets[k + 1])
                dec.write(es[int(pic_offsets[k]):end])
            pull()
        info = dict(width=dec.width, height=dec.height, coded_size=dec.coded_size, frame_rate=dec.frame_rate)
    return frames, indices, info


# ---------------------------------------------------------------------------------------------------------
# MP2 audio: the reference's 10-function decoder ABI (reference src/wasm/mp2.h:10-20), exported by the same
# three libraries.

_MP2_SIGS = {
    "mp2_decoder_create": (ctypes.c_void_p, [ctypes.c_uint, ctypes.c_int]),
    "mp2_decoder_destroy": (None, [ctypes.c_void_p]),
    "mp2_decoder_get_write_ptr": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_uint]),
    "mp2_decoder_get_index": (ctypes.c_int, [ctypes.c_void_p]),
    "mp2_decoder_set_index": (None, [ctypes.c_void_p, ctypes.c_uint]),
    "mp2_decoder_did_write": (None, [ctypes.c_void_p, ctypes.c_uint]),
    "mp2_decoder_get_left_channel_ptr": (ctypes.c_void_p, [ctypes.c_void_p]),
    "mp2_decoder_get_right_channel_ptr": (ctypes.c_void_p, [ctypes.c_void_p]),
    "mp2_decoder_get_sample_rate": (ctypes.c_int, [ctypes.c_void_p]),
    "mp2_decoder_decode": (ctypes.c_int, [ctypes.c_void_p]),
}
MP2_ABI_SYMBOLS = tuple(_MP2_SIGS)
MP2_SAMPLES_PER_FRAME = 1152

_mp2_libs = {}


def load_mp2(path):
    lib = _mp2_libs.get(path)
    if lib is None:
        lib = _load_shared(path)
        for name, (res, args) in _MP2_SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _mp2_libs[path] = lib
    return lib


class Mp2Decoder:
    """One MP2 decoder handle of whichever library `path` names."""

    def __init__(self, path, buffer_size=128 * 1024, mode=MODE_EXPAND):
        self.lib = load_mp2(path)
        self.h = self.lib.mp2_decoder_create(buffer_size, mode)
        if not self.h:
            raise RuntimeError("mp2_decoder_create failed (%s)" % path)

    def close(self):
        if self.h:
            self.lib.mp2_decoder_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def write(self, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n = int(data.size)
        ptr = self.lib.mp2_decoder_get_write_ptr(self.h, n)
        ctypes.memmove(ptr, data.ctypes.data, n)
        self.lib.mp2_decoder_did_write(self.h, n)

    def decode(self):
        """Bytes of the frame that was decoded, 0 if none."""
        return int(self.lib.mp2_decoder_decode(self.h))

    @property
    def index(self):
        return self.lib.mp2_decoder_get_index(self.h)

    @index.setter
    def index(self, v):
        self.lib.mp2_decoder_set_index(self.h, v)

    @property
    def sample_rate(self):
        return self.lib.mp2_decoder_get_sample_rate(self.h)

    def channels(self):
        """Copies of the most recently decoded (left, right) float32[1152]."""
        out = []
        for getter in (self.lib.mp2_decoder_get_left_channel_ptr, self.lib.mp2_decoder_get_right_channel_ptr):
            ptr = getter(self.h)
            out.append(np.ctypeslib.as_array((ctypes.c_float * MP2_SAMPLES_PER_FRAME).from_address(ptr)).copy())
        return tuple(out)


def decode_mp2_stream(path, data, frame_offsets=None, buffer_size=None, mode=MODE_EXPAND, max_frames=None):
    """Feeds `data` (one write, or one write per frame when `frame_offsets` is given, the way ts.js hands over PES
    payloads) and pulls every frame.  Returns (pcm float32[n_frames, 2, 1152], [bit index after each decode()],
    [frame bytes], sample rate after the last frame)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    pcm, indices, sizes = [], [], []
    with Mp2Decoder(path, buffer_size or (len(data) + 1024), mode) as dec:
        def pull():
            while max_frames is None or len(pcm) < max_frames:
                n = dec.decode()
                if n == 0:
                    break
                sizes.append(n)
                indices.append(dec.index)
                pcm.append(np.stack(dec.channels()))
        if frame_offsets is None:
            dec.write(data)
            pull()
        else:
            for k in range(len(frame_offsets) - 1):
                dec.write(data[int(frame_offsets[k]):int(frame_offsets[k + 1])])
                pull()
        rate = dec.sample_rate
    return (np.stack(pcm) if pcm else np.zeros((0, 2, MP2_SAMPLES_PER_FRAME), np.float32)), indices, sizes, rate
