"""ctypes front-end of the MP2 batch interface of include/jsmpeg_hip.h (part 3): many MPEG-1 Audio Layer II
streams, every frame, PCM left in HBM -- and of the live audio streams of part 6 (Mp2Live: streams that go on, a tick decodes
what has arrived of every one of them).  Host-side plumbing only; the decode happens in libjsmpeg_hip.so on the
GPU.  Loading fails loudly when the library is missing (there is no CPU decode in the product).

The reference's one-frame-per-call MP2 decoder ABI (mp2_decoder_*) is driven through jsmpeg_amd.cabi.Mp2Decoder."""
import ctypes
import os

import numpy as np

from . import build as _build

SAMPLES_PER_FRAME = 1152

MP2_BATCH_SYMBOLS = ("jsmpeg_hip_mp2_batch_create", "jsmpeg_hip_mp2_batch_destroy", "jsmpeg_hip_mp2_batch_upload",
                     "jsmpeg_hip_mp2_batch_decode", "jsmpeg_hip_mp2_batch_sync", "jsmpeg_hip_mp2_batch_frame_count",
                     "jsmpeg_hip_mp2_batch_frame_info", "jsmpeg_hip_mp2_batch_pcm", "jsmpeg_hip_mp2_batch_read_pcm",
                     "jsmpeg_hip_mp2_batch_timings", "jsmpeg_hip_mp2_batch_upload_ts", "jsmpeg_hip_mp2_batch_ts_writes",
                     "jsmpeg_hip_mp2_batch_read_bytes", "jsmpeg_hip_mp2_batch_upload_device")

MP2_LIVE_SYMBOLS = ("jsmpeg_hip_mp2_live_create", "jsmpeg_hip_mp2_live_destroy", "jsmpeg_hip_mp2_live_open", "jsmpeg_hip_mp2_live_close",
                    "jsmpeg_hip_mp2_live_write", "jsmpeg_hip_mp2_live_write_v", "jsmpeg_hip_mp2_live_write_ts", "jsmpeg_hip_mp2_live_tick",
                    "jsmpeg_hip_mp2_live_frame_count", "jsmpeg_hip_mp2_live_frame", "jsmpeg_hip_mp2_live_read_pcm",
                    "jsmpeg_hip_mp2_live_stream_info", "jsmpeg_hip_mp2_live_timings")


class LiveConfig(ctypes.Structure):            # jsmpeg_hip_mp2_live_config_t
    _fields_ = [("max_streams", ctypes.c_uint32), ("max_frames_per_tick", ctypes.c_uint32), ("store_bytes", ctypes.c_uint32),
                ("device", ctypes.c_int32)]


class LiveFrame(ctypes.Structure):             # jsmpeg_hip_mp2_live_frame_t
    _fields_ = [("stream", ctypes.c_uint32), ("sample_rate", ctypes.c_int32), ("pts", ctypes.c_double), ("stream_offset", ctypes.c_uint64),
                ("bytes", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("device_pcm", ctypes.c_void_p)]


class LiveStreamInfo(ctypes.Structure):        # jsmpeg_hip_mp2_live_stream_info_t
    _fields_ = [("sample_rate", ctypes.c_int32), ("pending_bytes", ctypes.c_uint32), ("bytes_written", ctypes.c_uint64),
                ("frames", ctypes.c_uint64), ("evictions", ctypes.c_uint64), ("stalled", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.LIB_HIP
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: build it with `python -m jsmpeg_amd.build hip` "
                               "(there is no CPU fallback for the MP2 decode stage)" % path)
        L = _build.load_hip_library(path)
        vp, u32, u64, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int32
        L.jsmpeg_hip_mp2_batch_create.restype = vp
        L.jsmpeg_hip_mp2_batch_create.argtypes = [u32, u64, i32]
        L.jsmpeg_hip_mp2_batch_destroy.restype = None
        L.jsmpeg_hip_mp2_batch_destroy.argtypes = [vp]
        L.jsmpeg_hip_mp2_batch_upload.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_batch_upload.argtypes = [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(u64)]
        L.jsmpeg_hip_mp2_batch_decode.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_batch_decode.argtypes = [vp, vp]
        L.jsmpeg_hip_mp2_batch_sync.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_batch_sync.argtypes = [vp]
        L.jsmpeg_hip_mp2_batch_frame_count.restype = u32
        L.jsmpeg_hip_mp2_batch_frame_count.argtypes = [vp, i32]
        L.jsmpeg_hip_mp2_batch_frame_info.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_batch_frame_info.argtypes = [vp, u32, u32, ctypes.POINTER(u32), ctypes.POINTER(u32),
                                                      ctypes.POINTER(i32)]
        L.jsmpeg_hip_mp2_batch_pcm.restype = vp
        L.jsmpeg_hip_mp2_batch_pcm.argtypes = [vp]
        L.jsmpeg_hip_mp2_batch_read_pcm.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_batch_read_pcm.argtypes = [vp, u32, u32, u32, vp]
        L.jsmpeg_hip_mp2_batch_timings.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_batch_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        L.jsmpeg_hip_mp2_batch_upload_ts.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_batch_upload_ts.argtypes = [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(u64), u32]
        L.jsmpeg_hip_mp2_batch_ts_writes.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_batch_ts_writes.argtypes = [vp, u32, vp, vp, vp, u32]
        L.jsmpeg_hip_mp2_batch_read_bytes.restype = ctypes.c_int64
        L.jsmpeg_hip_mp2_batch_read_bytes.argtypes = [vp, u32, vp, u64]
        L.jsmpeg_hip_mp2_batch_upload_device.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_batch_upload_device.argtypes = [vp, vp, u64, u32, vp, vp, vp]
        L.jsmpeg_hip_mp2_live_create.restype = vp
        L.jsmpeg_hip_mp2_live_create.argtypes = [ctypes.POINTER(LiveConfig)]
        L.jsmpeg_hip_mp2_live_destroy.restype = None
        L.jsmpeg_hip_mp2_live_destroy.argtypes = [vp]
        L.jsmpeg_hip_mp2_live_open.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_live_open.argtypes = [vp]
        L.jsmpeg_hip_mp2_live_close.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_live_close.argtypes = [vp, u32]
        L.jsmpeg_hip_mp2_live_write.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_live_write.argtypes = [vp, u32, ctypes.c_double, vp, u32]
        L.jsmpeg_hip_mp2_live_write_v.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_live_write_v.argtypes = [vp, u32, ctypes.c_double, ctypes.POINTER(vp), ctypes.POINTER(u32), u32]
        L.jsmpeg_hip_mp2_live_write_ts.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_live_write_ts.argtypes = [vp, u32, vp, u32, u32]
        L.jsmpeg_hip_mp2_live_tick.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_live_tick.argtypes = [vp, vp]
        L.jsmpeg_hip_mp2_live_frame_count.restype = u32
        L.jsmpeg_hip_mp2_live_frame_count.argtypes = [vp]
        L.jsmpeg_hip_mp2_live_frame.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_live_frame.argtypes = [vp, u32, ctypes.POINTER(LiveFrame)]
        L.jsmpeg_hip_mp2_live_read_pcm.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_live_read_pcm.argtypes = [vp, u32, u32, vp]
        L.jsmpeg_hip_mp2_live_stream_info.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_live_stream_info.argtypes = [vp, u32, ctypes.POINTER(LiveStreamInfo)]
        L.jsmpeg_hip_mp2_live_timings.restype = ctypes.c_int
        L.jsmpeg_hip_mp2_live_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        L.jsmpeg_hip_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _err():
    return (lib().jsmpeg_hip_last_error() or b"").decode()


class Mp2Batch:
    """One batch decoder: upload N streams, decode, read PCM (or take the device pointer)."""

    def __init__(self, max_streams, max_bytes, device=-1):
        self.L = lib()
        self.h = self.L.jsmpeg_hip_mp2_batch_create(max_streams, max_bytes, device)
        if not self.h:
            raise RuntimeError("jsmpeg_hip_mp2_batch_create failed: " + _err())
        self.n_streams = 0

    def close(self):
        if self.h:
            self.L.jsmpeg_hip_mp2_batch_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload(self, streams):
        streams = [np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
        n = len(streams)
        ptrs = (ctypes.c_void_p * n)(*[s.ctypes.data for s in streams])
        lens = (ctypes.c_uint64 * n)(*[s.size for s in streams])
        if self.L.jsmpeg_hip_mp2_batch_upload(self.h, n, ptrs, lens) < 0:
            raise RuntimeError("jsmpeg_hip_mp2_batch_upload failed: " + _err())
        self.n_streams = n

    def upload_device(self, dev_ptr, total_bytes, begin, end, hip_stream=None):
        """Streams already in device memory: one packed buffer, byte ranges [begin[i], end[i])."""
        begin = np.ascontiguousarray(begin, dtype=np.uint32)
        end = np.ascontiguousarray(end, dtype=np.uint32)
        if self.L.jsmpeg_hip_mp2_batch_upload_device(self.h, dev_ptr, total_bytes, len(begin), begin.ctypes.data, end.ctypes.data,
                                                     hip_stream) < 0:
            raise RuntimeError("jsmpeg_hip_mp2_batch_upload_device failed: " + _err())
        self.n_streams = len(begin)

    def upload_ts(self, ts_buffers, stream_id=0xC0):
        """MPEG-TS buffers in; the audio stream's payload is demultiplexed on the device (reference ts.js semantics)."""
        bufs = [np.ascontiguousarray(s, dtype=np.uint8) for s in ts_buffers]
        n = len(bufs)
        ptrs = (ctypes.c_void_p * n)(*[s.ctypes.data for s in bufs])
        lens = (ctypes.c_uint64 * n)(*[s.size for s in bufs])
        if self.L.jsmpeg_hip_mp2_batch_upload_ts(self.h, n, ptrs, lens, stream_id) < 0:
            raise RuntimeError("jsmpeg_hip_mp2_batch_upload_ts failed: " + _err())
        self.n_streams = n

    def ts_writes(self, stream):
        """[(pts seconds, offset, length)] -- the destination.write calls ts.js would have made."""
        n = self.L.jsmpeg_hip_mp2_batch_ts_writes(self.h, stream, None, None, None, 0)
        if n < 0:
            raise RuntimeError("jsmpeg_hip_mp2_batch_ts_writes failed: " + _err())
        pts = np.zeros(n, np.float64); off = np.zeros(n, np.uint32); ln = np.zeros(n, np.uint32)
        if n:
            self.L.jsmpeg_hip_mp2_batch_ts_writes(self.h, stream, pts.ctypes.data, off.ctypes.data, ln.ctypes.data, n)
        return [(float(pts[i]), int(off[i]), int(ln[i])) for i in range(n)]

    def read_bytes(self, stream):
        n = self.L.jsmpeg_hip_mp2_batch_read_bytes(self.h, stream, None, 0)
        if n < 0:
            raise RuntimeError("jsmpeg_hip_mp2_batch_read_bytes failed: " + _err())
        out = np.zeros(int(n), np.uint8)
        if n:
            self.L.jsmpeg_hip_mp2_batch_read_bytes(self.h, stream, out.ctypes.data, int(n))
        return out

    def decode(self, hip_stream=None, sync=True):
        n = self.L.jsmpeg_hip_mp2_batch_decode(self.h, hip_stream)
        if n < 0:
            raise RuntimeError("jsmpeg_hip_mp2_batch_decode failed: " + _err())
        if sync and self.L.jsmpeg_hip_mp2_batch_sync(self.h) < 0:
            raise RuntimeError("jsmpeg_hip_mp2_batch_sync failed: " + _err())
        return n

    def frame_count(self, stream=-1):
        return int(self.L.jsmpeg_hip_mp2_batch_frame_count(self.h, stream))

    def frame_info(self, stream, frame):
        off, size, rate = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_int32()
        if self.L.jsmpeg_hip_mp2_batch_frame_info(self.h, stream, frame, ctypes.byref(off), ctypes.byref(size),
                                                  ctypes.byref(rate)) < 0:
            raise RuntimeError("jsmpeg_hip_mp2_batch_frame_info failed: " + _err())
        return off.value, size.value, rate.value

    def pcm_device_pointer(self):
        return self.L.jsmpeg_hip_mp2_batch_pcm(self.h)

    def read_pcm(self, stream, first=0, count=None):
        """float32[count, 2, 1152] (left, right) of one stream."""
        have = self.frame_count(stream)
        if count is None:
            count = have - first
        out = np.empty((count, 2, SAMPLES_PER_FRAME), dtype=np.float32)
        if self.L.jsmpeg_hip_mp2_batch_read_pcm(self.h, stream, first, count, out.ctypes.data) < 0:
            raise RuntimeError("jsmpeg_hip_mp2_batch_read_pcm failed: " + _err())
        return out

    def timings(self):
        t = (ctypes.c_float * 5)()
        if self.L.jsmpeg_hip_mp2_batch_timings(self.h, t) < 0:
            raise RuntimeError("jsmpeg_hip_mp2_batch_timings failed: " + _err())
        return dict(zip(("walk_ms", "side_ms", "matrix_ms", "window_ms", "total_ms"), [float(x) for x in t]))


class Mp2Live:
    """Live audio streams (include/jsmpeg_hip.h part 6): open() streams, write(stream, pts, bytes) what arrives, tick() decodes
    every completely buffered frame of every stream in one pass; the samples stay in HBM (frames()[i]["device_pcm"]) or come to
    the host with read_pcm().  Per stream what the reference's MP2 decoder gives for the same write() calls."""

    def __init__(self, max_streams, max_frames_per_tick=0, store_bytes=0, device=-1):
        self.L = lib()
        cfg = LiveConfig(max_streams, max_frames_per_tick, store_bytes, device)
        self.h = self.L.jsmpeg_hip_mp2_live_create(ctypes.byref(cfg))
        if not self.h:
            raise RuntimeError("jsmpeg_hip_mp2_live_create failed: " + _err())

    def close(self):
        if self.h:
            self.L.jsmpeg_hip_mp2_live_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def open(self):
        s = self.L.jsmpeg_hip_mp2_live_open(self.h)
        if s < 0:
            raise RuntimeError("jsmpeg_hip_mp2_live_open failed: " + _err())
        return s

    def close_stream(self, stream):
        if self.L.jsmpeg_hip_mp2_live_close(self.h, stream) < 0:
            raise RuntimeError("jsmpeg_hip_mp2_live_close failed: " + _err())

    def write(self, stream, pts, data):
        """One write(pts, buffers) of the reference's decoder; `data`: bytes-like, or a list of them (ONE write of the total)."""
        if isinstance(data, (list, tuple)):
            bufs = [np.ascontiguousarray(np.frombuffer(d, np.uint8) if isinstance(d, (bytes, bytearray, memoryview)) else d, dtype=np.uint8) for d in data]
            n = len(bufs)
            ptrs = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bufs])
            lens = (ctypes.c_uint32 * n)(*[b.size for b in bufs])
            rc = self.L.jsmpeg_hip_mp2_live_write_v(self.h, stream, pts, ptrs, lens, n)
        else:
            b = np.ascontiguousarray(np.frombuffer(data, np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else data, dtype=np.uint8)
            rc = self.L.jsmpeg_hip_mp2_live_write(self.h, stream, pts, b.ctypes.data, b.size)
        if rc < 0:
            raise RuntimeError("jsmpeg_hip_mp2_live_write failed: " + _err())

    def write_ts(self, stream, data, stream_id=0xC0):
        b = np.ascontiguousarray(np.frombuffer(data, np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else data, dtype=np.uint8)
        if self.L.jsmpeg_hip_mp2_live_write_ts(self.h, stream, b.ctypes.data, b.size, stream_id) < 0:
            raise RuntimeError("jsmpeg_hip_mp2_live_write_ts failed: " + _err())

    def tick(self, hip_stream=None):
        n = self.L.jsmpeg_hip_mp2_live_tick(self.h, hip_stream)
        if n < 0:
            raise RuntimeError("jsmpeg_hip_mp2_live_tick failed: " + _err())
        return n

    def frames(self):
        """The last tick's frames: [{stream, sample_rate, pts, stream_offset, bytes, device_pcm}]"""
        out = []
        f = LiveFrame()
        for i in range(self.L.jsmpeg_hip_mp2_live_frame_count(self.h)):
            if self.L.jsmpeg_hip_mp2_live_frame(self.h, i, ctypes.byref(f)) < 0:
                raise RuntimeError("jsmpeg_hip_mp2_live_frame failed: " + _err())
            out.append({"stream": f.stream, "sample_rate": f.sample_rate, "pts": f.pts, "stream_offset": f.stream_offset, "bytes": f.bytes,
                        "device_pcm": f.device_pcm})
        return out

    def read_pcm(self, first=0, count=None):
        """float32[count, 2, 1152] (left, right) of frames first .. of the last tick."""
        have = self.L.jsmpeg_hip_mp2_live_frame_count(self.h)
        if count is None:
            count = have - first
        out = np.empty((count, 2, SAMPLES_PER_FRAME), dtype=np.float32)
        if self.L.jsmpeg_hip_mp2_live_read_pcm(self.h, first, count, out.ctypes.data) < 0:
            raise RuntimeError("jsmpeg_hip_mp2_live_read_pcm failed: " + _err())
        return out

    def stream_info(self, stream):
        i = LiveStreamInfo()
        if self.L.jsmpeg_hip_mp2_live_stream_info(self.h, stream, ctypes.byref(i)) < 0:
            raise RuntimeError("jsmpeg_hip_mp2_live_stream_info failed: " + _err())
        return {k: getattr(i, k) for k, _ in LiveStreamInfo._fields_ if k != "reserved"}

    def timings(self):
        t = (ctypes.c_float * 7)()
        if self.L.jsmpeg_hip_mp2_live_timings(self.h, t) < 0:
            raise RuntimeError("jsmpeg_hip_mp2_live_timings failed: " + _err())
        return dict(zip(("enqueue_ms", "wait_ms", "book_ms", "total_ms", "walk_ms", "matrix_ms", "window_ms"), [float(x) for x in t]))
