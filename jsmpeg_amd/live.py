"""ctypes front-end of the live-stream interface of include/jsmpeg_hip.h (part 5): N streams that go on, every pending
picture of every stream in one pass of the batch engine per tick.  Host-side plumbing only (tests, bench.py); the Node
host is jsmpeg_amd/js/live-hip.js over the same C ABI.  Loading fails loudly when the library is missing."""
import ctypes

import numpy as np

from . import batch as _batch

FLUSH = 1

LIVE_SYMBOLS = ("jsmpeg_hip_live_create", "jsmpeg_hip_live_destroy", "jsmpeg_hip_live_open", "jsmpeg_hip_live_close",
                "jsmpeg_hip_live_write", "jsmpeg_hip_live_write_v", "jsmpeg_hip_live_write_ts", "jsmpeg_hip_live_tick", "jsmpeg_hip_live_tick_begin", "jsmpeg_hip_live_tick_end", "jsmpeg_hip_live_picture_count", "jsmpeg_hip_live_picture",
                "jsmpeg_hip_live_geometry", "jsmpeg_hip_live_read_frame", "jsmpeg_hip_live_read_frames", "jsmpeg_hip_live_read_frames_begin", "jsmpeg_hip_live_read_frames_end", "jsmpeg_hip_live_read_rgba",
                "jsmpeg_hip_host_alloc", "jsmpeg_hip_host_free", "jsmpeg_hip_host_register", "jsmpeg_hip_host_unregister",
                "jsmpeg_hip_live_frame_hashes", "jsmpeg_hip_live_stream_info", "jsmpeg_hip_live_timings")


class LiveConfig(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("max_streams", ctypes.c_uint32),
                ("max_pictures_per_tick", ctypes.c_uint32), ("store_bytes", ctypes.c_uint32), ("device", ctypes.c_int32)]


class LivePicture(ctypes.Structure):
    _fields_ = [("stream", ctypes.c_uint32), ("type", ctypes.c_int32), ("pts", ctypes.c_double),
                ("stream_offset", ctypes.c_uint64), ("device_frame", ctypes.c_void_p)]


class LiveStreamInfo(ctypes.Structure):
    _fields_ = [("has_sequence_header", ctypes.c_int32), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("frame_rate", ctypes.c_float), ("status", ctypes.c_int32), ("pending_bytes", ctypes.c_uint32),
                ("bytes_written", ctypes.c_uint64), ("pictures", ctypes.c_uint64), ("evictions", ctypes.c_uint64)]


_bound = False


def lib():
    global _bound
    L = _batch.lib()
    if not _bound:
        vp, u32, u64, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int32
        L.jsmpeg_hip_live_create.restype = vp
        L.jsmpeg_hip_live_create.argtypes = [ctypes.POINTER(LiveConfig)]
        L.jsmpeg_hip_live_destroy.restype = None
        L.jsmpeg_hip_live_destroy.argtypes = [vp]
        L.jsmpeg_hip_live_open.restype = ctypes.c_int
        L.jsmpeg_hip_live_open.argtypes = [vp]
        L.jsmpeg_hip_live_close.restype = ctypes.c_int
        L.jsmpeg_hip_live_close.argtypes = [vp, u32]
        L.jsmpeg_hip_live_write.restype = ctypes.c_int
        L.jsmpeg_hip_live_write.argtypes = [vp, u32, ctypes.c_double, vp, u32]
        L.jsmpeg_hip_live_write_ts.restype = ctypes.c_int
        L.jsmpeg_hip_live_write_ts.argtypes = [vp, u32, vp, u32, u32]
        L.jsmpeg_hip_live_tick.restype = ctypes.c_int
        L.jsmpeg_hip_live_tick.argtypes = [vp, u32, vp]
        L.jsmpeg_hip_live_tick_begin.restype = ctypes.c_int
        L.jsmpeg_hip_live_tick_begin.argtypes = [vp, u32, vp]
        L.jsmpeg_hip_live_tick_end.restype = ctypes.c_int
        L.jsmpeg_hip_live_tick_end.argtypes = [vp]
        L.jsmpeg_hip_live_picture_count.restype = u32
        L.jsmpeg_hip_live_picture_count.argtypes = [vp]
        L.jsmpeg_hip_live_picture.restype = ctypes.c_int
        L.jsmpeg_hip_live_picture.argtypes = [vp, u32, ctypes.POINTER(LivePicture)]
        L.jsmpeg_hip_live_geometry.restype = ctypes.c_int
        L.jsmpeg_hip_live_geometry.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(u32), ctypes.POINTER(u32)]
        L.jsmpeg_hip_live_read_frame.restype = ctypes.c_int
        L.jsmpeg_hip_live_read_frame.argtypes = [vp, u32, vp, vp, vp]
        L.jsmpeg_hip_live_read_frames.restype = ctypes.c_int
        L.jsmpeg_hip_live_read_frames.argtypes = [vp, u32, u32, vp, ctypes.c_uint64]
        L.jsmpeg_hip_live_read_frames_begin.restype = ctypes.c_int
        L.jsmpeg_hip_live_read_frames_begin.argtypes = [vp, u32, u32, vp, ctypes.c_uint64]
        L.jsmpeg_hip_live_read_frames_end.restype = ctypes.c_int
        L.jsmpeg_hip_live_read_frames_end.argtypes = [vp]
        L.jsmpeg_hip_host_alloc.restype = vp
        L.jsmpeg_hip_host_alloc.argtypes = [ctypes.c_uint64]
        L.jsmpeg_hip_host_free.restype = None
        L.jsmpeg_hip_host_free.argtypes = [vp]
        L.jsmpeg_hip_host_register.restype = ctypes.c_int
        L.jsmpeg_hip_host_register.argtypes = [vp, ctypes.c_uint64]
        L.jsmpeg_hip_host_unregister.restype = ctypes.c_int
        L.jsmpeg_hip_host_unregister.argtypes = [vp]
        L.jsmpeg_hip_live_read_rgba.restype = ctypes.c_int
        L.jsmpeg_hip_live_read_rgba.argtypes = [vp, u32, vp]
        L.jsmpeg_hip_live_frame_hashes.restype = ctypes.c_int
        L.jsmpeg_hip_live_frame_hashes.argtypes = [vp, vp]
        L.jsmpeg_hip_live_stream_info.restype = ctypes.c_int
        L.jsmpeg_hip_live_stream_info.argtypes = [vp, u32, ctypes.POINTER(LiveStreamInfo)]
        L.jsmpeg_hip_live_timings.restype = ctypes.c_int
        L.jsmpeg_hip_live_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        _bound = True
    return L


class Live:
    """N live streams -> per tick, every pending picture's Y/Cr/Cb planes in HBM."""

    def __init__(self, width, height, max_streams, pictures_per_tick=0, store_bytes=0, device=-1):
        self.L = lib()
        cfg = LiveConfig(width, height, max_streams, pictures_per_tick, store_bytes, device)
        self.width, self.height = width, height
        self.h = self.L.jsmpeg_hip_live_create(ctypes.byref(cfg))
        if not self.h:
            raise RuntimeError("jsmpeg_hip_live_create: " + _batch.last_error())
        cw, ch, lu, chb = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_uint32(), ctypes.c_uint32()
        self._ok(self.L.jsmpeg_hip_live_geometry(self.h, cw, ch, lu, chb))
        self.coded_width, self.coded_height, self.luma_bytes, self.chroma_bytes = cw.value, ch.value, lu.value, chb.value

    def _ok(self, rc):
        if rc < 0:
            raise RuntimeError(_batch.last_error())
        return rc

    def close(self):
        if getattr(self, "_pin", None):
            self.L.jsmpeg_hip_host_free(self._pin)
            self._pin, self._pin_bytes = None, 0
        if self.h:
            self.L.jsmpeg_hip_live_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def open(self):
        return self._ok(self.L.jsmpeg_hip_live_open(self.h))

    def close_stream(self, stream):
        self._ok(self.L.jsmpeg_hip_live_close(self.h, stream))

    def write(self, stream, data, pts=0.0):
        a = np.ascontiguousarray(data, dtype=np.uint8)
        self._ok(self.L.jsmpeg_hip_live_write(self.h, stream, pts, a.ctypes.data, a.size))

    def write_ts(self, stream, data, stream_id=0xE0):
        """MPEG-TS bytes in any pieces: the reference's demuxer in front of write(), its state kept per stream"""
        a = np.ascontiguousarray(data, dtype=np.uint8)
        self._ok(self.L.jsmpeg_hip_live_write_ts(self.h, stream, a.ctypes.data, a.size, stream_id))

    def tick(self, flush=True, stream=None):
        """one pass over what has been written; returns the pictures decoded (see pictures())"""
        return self._ok(self.L.jsmpeg_hip_live_tick(self.h, FLUSH if flush else 0, stream))

    def tick_begin(self, flush=True, stream=None):
        """the tick's first half: the pass is on the device when this returns; write() / write_ts() may go on meanwhile"""
        self._ok(self.L.jsmpeg_hip_live_tick_begin(self.h, FLUSH if flush else 0, stream))

    def tick_end(self):
        """the second half: waits for the pass; returns the pictures decoded (the writes made meanwhile are behind it)"""
        return self._ok(self.L.jsmpeg_hip_live_tick_end(self.h))

    @property
    def picture_count(self):
        return self.L.jsmpeg_hip_live_picture_count(self.h)

    def pictures(self):
        out = []
        for i in range(self.picture_count):
            p = LivePicture()
            self._ok(self.L.jsmpeg_hip_live_picture(self.h, i, ctypes.byref(p)))
            out.append(p)
        return out

    def read_frame(self, i):
        y = np.empty(self.luma_bytes, dtype=np.uint8)
        cr = np.empty(self.chroma_bytes, dtype=np.uint8)
        cb = np.empty(self.chroma_bytes, dtype=np.uint8)
        self._ok(self.L.jsmpeg_hip_live_read_frame(self.h, i, y.ctypes.data, cr.ctypes.data, cb.ctypes.data))
        return y, cr, cb

    def read_frames(self, first=0, count=None):
        """every picture of the last tick in one go, into pinned memory of the object's own (grown on demand): an array
        [count, luma_bytes + 2 * chroma_bytes] -- row k is Y | Cr | Cb of picture first + k -- valid until the next call"""
        n = self.picture_count - first if count is None else count
        planes = self.luma_bytes + 2 * self.chroma_bytes
        if n <= 0:
            return np.empty((0, planes), dtype=np.uint8)
        if getattr(self, "_pin_bytes", 0) < n * planes:
            if getattr(self, "_pin", None):
                self.L.jsmpeg_hip_host_free(self._pin)
            self._pin_bytes = max(n, 2 * getattr(self, "_pin_pictures", 0)) * planes
            self._pin_pictures = self._pin_bytes // planes
            self._pin = self.L.jsmpeg_hip_host_alloc(self._pin_bytes)
            if not self._pin:
                self._pin_bytes = 0
                raise RuntimeError("jsmpeg_hip_host_alloc: " + _batch.last_error())
        self._ok(self.L.jsmpeg_hip_live_read_frames(self.h, first, n, self._pin, planes))
        return np.ctypeslib.as_array((ctypes.c_uint8 * (n * planes)).from_address(self._pin)).reshape(n, planes)

    def read_frames_begin(self, first=0, count=None):
        """the read-out in two halves: the last tick's pictures start on their way to pinned memory of the object's own (two
        buffers in turn) and the call returns; writes and the NEXT tick go on beside the copies; read_frames_end() waits and
        hands out the array (valid until the read-out after the next one)"""
        n = self.picture_count - first if count is None else count
        planes = self.luma_bytes + 2 * self.chroma_bytes
        if not hasattr(self, "_pins"):
            self._pins, self._pin_turn, self._pending = [[None, 0], [None, 0]], 0, None
        if self._pending is not None:
            raise RuntimeError("read_frames_begin: a read-out is in flight (read_frames_end first)")
        slot = self._pins[self._pin_turn]
        if slot[1] < max(1, n) * planes:
            if slot[0]:
                self.L.jsmpeg_hip_host_free(slot[0])
            slot[1] = max(1, 2 * n) * planes
            slot[0] = self.L.jsmpeg_hip_host_alloc(slot[1])
            if not slot[0]:
                slot[1] = 0
                raise RuntimeError("jsmpeg_hip_host_alloc: " + _batch.last_error())
        self._ok(self.L.jsmpeg_hip_live_read_frames_begin(self.h, first, max(0, n), slot[0], planes))
        self._pending = (slot[0], max(0, n), planes)
        self._pin_turn ^= 1

    def read_frames_end(self):
        if getattr(self, "_pending", None) is None:
            return None
        ptr, n, planes = self._pending
        self._pending = None
        self._ok(self.L.jsmpeg_hip_live_read_frames_end(self.h))
        if n == 0:
            return np.empty((0, planes), dtype=np.uint8)
        return np.ctypeslib.as_array((ctypes.c_uint8 * (n * planes)).from_address(ptr)).reshape(n, planes)

    def read_rgba(self, i):
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._ok(self.L.jsmpeg_hip_live_read_rgba(self.h, i, out.ctypes.data))
        return out

    def frame_hashes(self):
        n = self.picture_count
        out = np.zeros(max(1, n), dtype=np.uint64)
        self._ok(self.L.jsmpeg_hip_live_frame_hashes(self.h, out.ctypes.data))
        return out[:n]

    def stream_info(self, stream):
        info = LiveStreamInfo()
        self._ok(self.L.jsmpeg_hip_live_stream_info(self.h, stream, ctypes.byref(info)))
        return info

    def timings(self):
        ms = (ctypes.c_float * 9)()
        self._ok(self.L.jsmpeg_hip_live_timings(self.h, ms))
        keys = ("stage_ms", "decode_call_ms", "wait_ms", "book_ms", "total_ms", "index_ms", "host_ms", "parse_ms", "recon_ms")
        return dict(zip(keys, [float(x) for x in ms]))
