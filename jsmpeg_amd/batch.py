"""ctypes front-end of the batch interface of include/jsmpeg_hip.h (part 2).
Host-side plumbing only; every byte of decode work happens in libjsmpeg_hip.so
on the GPU.  Loading fails loudly when the library is missing."""
import ctypes
import os

import numpy as np

from . import build as _build


class BatchConfig(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("max_streams", ctypes.c_uint32),
                ("max_pictures", ctypes.c_uint32), ("max_es_bytes", ctypes.c_uint64), ("device", ctypes.c_int32)]


class PictureInfo(ctypes.Structure):
    _fields_ = [("stream", ctypes.c_uint32), ("es_offset", ctypes.c_uint32), ("type", ctypes.c_int32),
                ("decoded", ctypes.c_int32), ("level", ctypes.c_int32), ("forward", ctypes.c_int32),
                ("n_slices", ctypes.c_uint32)]


BATCH_SYMBOLS = ("jsmpeg_hip_batch_create", "jsmpeg_hip_batch_destroy", "jsmpeg_hip_batch_upload",
                 "jsmpeg_hip_batch_upload_device", "jsmpeg_hip_batch_attach_device", "jsmpeg_hip_batch_decode", "jsmpeg_hip_batch_sync",
                 "jsmpeg_hip_batch_picture_count", "jsmpeg_hip_batch_picture_info", "jsmpeg_hip_batch_geometry",
                 "jsmpeg_hip_batch_frame_pool", "jsmpeg_hip_batch_read_frame", "jsmpeg_hip_batch_read_frames", "jsmpeg_hip_batch_frame_hashes",
                 "jsmpeg_hip_batch_timings", "jsmpeg_hip_batch_level_timings", "jsmpeg_hip_batch_counters", "jsmpeg_hip_batch_recon_info", "jsmpeg_hip_batch_link_streams", "jsmpeg_hip_batch_seed_stream", "jsmpeg_hip_batch_uncovered", "jsmpeg_hip_batch_render_rgba",
                 "jsmpeg_hip_batch_read_rgba", "jsmpeg_hip_batch_render_rgba_gl", "jsmpeg_hip_batch_read_rgba_gl", "jsmpeg_hip_batch_upload_ts", "jsmpeg_hip_batch_upload_ts_writes", "jsmpeg_hip_batch_ts_writes",
                 "jsmpeg_hip_batch_read_es", "jsmpeg_hip_batch_stream_info",
                 "jsmpeg_hip_decoder_render_rgba", "jsmpeg_hip_last_error",
                 "jsmpeg_hip_device_count", "jsmpeg_hip_decoder_get_device_frame", "jsmpeg_hip_decoder_ahead_stats")

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.LIB_HIP
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: build it with `python -m jsmpeg_amd.build hip` "
                               "(there is no CPU fallback for the decode path)" % path)
        L = _build.load_hip_library(path)
        vp, u32, u64, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int32
        L.jsmpeg_hip_batch_create.restype = vp
        L.jsmpeg_hip_batch_create.argtypes = [ctypes.POINTER(BatchConfig)]
        L.jsmpeg_hip_batch_destroy.restype = None
        L.jsmpeg_hip_batch_destroy.argtypes = [vp]
        L.jsmpeg_hip_batch_upload.restype = ctypes.c_int
        L.jsmpeg_hip_batch_upload.argtypes = [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(u64)]
        L.jsmpeg_hip_batch_upload_device.restype = ctypes.c_int
        L.jsmpeg_hip_batch_upload_device.argtypes = [vp, vp, u64, u32, vp, vp, vp]
        L.jsmpeg_hip_batch_attach_device.restype = ctypes.c_int
        L.jsmpeg_hip_batch_attach_device.argtypes = [vp, vp, u64, u32, vp, vp, vp]
        L.jsmpeg_hip_batch_decode.restype = ctypes.c_int
        L.jsmpeg_hip_batch_decode.argtypes = [vp, vp]
        L.jsmpeg_hip_batch_sync.restype = ctypes.c_int
        L.jsmpeg_hip_batch_sync.argtypes = [vp]
        L.jsmpeg_hip_batch_picture_count.restype = u32
        L.jsmpeg_hip_batch_picture_count.argtypes = [vp]
        L.jsmpeg_hip_batch_picture_info.restype = ctypes.c_int
        L.jsmpeg_hip_batch_picture_info.argtypes = [vp, u32, ctypes.POINTER(PictureInfo)]
        L.jsmpeg_hip_batch_geometry.restype = ctypes.c_int
        L.jsmpeg_hip_batch_geometry.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(u32),
                                                ctypes.POINTER(u32), ctypes.POINTER(u64)]
        L.jsmpeg_hip_batch_frame_pool.restype = vp
        L.jsmpeg_hip_batch_frame_pool.argtypes = [vp]
        L.jsmpeg_hip_batch_read_frame.restype = ctypes.c_int
        L.jsmpeg_hip_batch_read_frame.argtypes = [vp, u32, vp, vp, vp]
        L.jsmpeg_hip_batch_read_frames.restype = ctypes.c_int
        L.jsmpeg_hip_batch_read_frames.argtypes = [vp, u32, u32, vp, ctypes.c_uint64]
        L.jsmpeg_hip_host_alloc.restype = vp
        L.jsmpeg_hip_host_alloc.argtypes = [ctypes.c_uint64]
        L.jsmpeg_hip_host_free.restype = None
        L.jsmpeg_hip_host_free.argtypes = [vp]
        L.jsmpeg_hip_batch_frame_hashes.restype = ctypes.c_int
        L.jsmpeg_hip_batch_frame_hashes.argtypes = [vp, vp]
        L.jsmpeg_hip_batch_render_rgba.restype = ctypes.c_int
        L.jsmpeg_hip_batch_render_rgba.argtypes = [vp, u32, u32, vp, vp]
        L.jsmpeg_hip_batch_upload_ts.restype = ctypes.c_int
        L.jsmpeg_hip_batch_upload_ts.argtypes = [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(u64), u32]
        L.jsmpeg_hip_batch_upload_ts_writes.restype = ctypes.c_int
        L.jsmpeg_hip_batch_upload_ts_writes.argtypes = [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(u64), ctypes.POINTER(u32),
                                                       ctypes.POINTER(u64), u32]
        L.jsmpeg_hip_ts_packet_runs.restype = ctypes.c_int
        L.jsmpeg_hip_ts_packet_runs.argtypes = [vp, u64, ctypes.POINTER(u64), u32, ctypes.POINTER(u64), ctypes.POINTER(u32), u32,
                                               ctypes.POINTER(u64), ctypes.POINTER(u64)]
        L.jsmpeg_hip_batch_ts_writes.restype = ctypes.c_int
        L.jsmpeg_hip_batch_ts_writes.argtypes = [vp, u32, vp, vp, vp, u32]
        L.jsmpeg_hip_batch_read_es.restype = ctypes.c_int64
        L.jsmpeg_hip_batch_read_es.argtypes = [vp, u32, vp, u64]
        L.jsmpeg_hip_batch_read_rgba.restype = ctypes.c_int
        L.jsmpeg_hip_batch_read_rgba.argtypes = [vp, u32, vp]
        L.jsmpeg_hip_batch_timings.restype = ctypes.c_int
        L.jsmpeg_hip_batch_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        L.jsmpeg_hip_batch_counters.restype = ctypes.c_int
        L.jsmpeg_hip_batch_counters.argtypes = [vp, ctypes.POINTER(u64)]
        L.jsmpeg_hip_last_error.restype = ctypes.c_char_p
        L.jsmpeg_hip_device_count.restype = ctypes.c_int
        _lib = L
    return _lib


def last_error():
    return lib().jsmpeg_hip_last_error().decode()


class Batch:
    """Many elementary streams -> every picture's Y/Cr/Cb planes in HBM."""

    def __init__(self, width, height, max_streams, max_pictures, max_es_bytes, device=-1):
        self.L = lib()
        cfg = BatchConfig(width, height, max_streams, max_pictures, max_es_bytes, device)
        self.width, self.height = width, height
        self.h = self.L.jsmpeg_hip_batch_create(ctypes.byref(cfg))
        if not self.h:
            raise RuntimeError("jsmpeg_hip_batch_create: " + last_error())
        cw, ch, lu, chb, fs = (ctypes.c_int32(), ctypes.c_int32(), ctypes.c_uint32(), ctypes.c_uint32(),
                               ctypes.c_uint64())
        self._ok(self.L.jsmpeg_hip_batch_geometry(self.h, cw, ch, lu, chb, fs))
        self.coded_width, self.coded_height = cw.value, ch.value
        self.luma_bytes, self.chroma_bytes, self.frame_stride = lu.value, chb.value, fs.value

    def _ok(self, rc):
        if rc < 0:
            raise RuntimeError(last_error())
        return rc

    def close(self):
        if getattr(self, "_pin", None):
            self.L.jsmpeg_hip_host_free(self._pin)
            self._pin, self._pin_bytes = None, 0
        if self.h:
            self.L.jsmpeg_hip_batch_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload(self, streams):
        arrs = [np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
        n = len(arrs)
        ptrs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs])
        lens = (ctypes.c_uint64 * n)(*[a.size for a in arrs])
        self._ok(self.L.jsmpeg_hip_batch_upload(self.h, n, ptrs, lens))

    def upload_ts(self, ts_buffers, stream_id=0xE0, write_sizes=None):
        """MPEG-TS in, demultiplexed on the device (reference src/ts.js semantics: resync, leftover bytes).  One write()
        per buffer, or -- write_sizes: per buffer a list of byte counts -- that buffer in several write() calls."""
        arrs = [np.ascontiguousarray(s, dtype=np.uint8) for s in ts_buffers]
        n = len(arrs)
        ptrs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs])
        lens = (ctypes.c_uint64 * n)(*[a.size for a in arrs])
        if write_sizes is None:
            self._ok(self.L.jsmpeg_hip_batch_upload_ts(self.h, n, ptrs, lens, stream_id))
            return
        counts = (ctypes.c_uint32 * n)(*[len(w) for w in write_sizes])
        flat = [int(x) for w in write_sizes for x in w]
        sizes = (ctypes.c_uint64 * max(1, len(flat)))(*flat)
        self._ok(self.L.jsmpeg_hip_batch_upload_ts_writes(self.h, n, ptrs, lens, counts, sizes, stream_id))

    def ts_writes(self, stream):
        """[(pts seconds, offset, length)]: the destination.write calls ts.js would have made for `stream`."""
        n = self._ok(self.L.jsmpeg_hip_batch_ts_writes(self.h, stream, None, None, None, 0))
        pts = np.zeros(max(1, n), dtype=np.float64)
        off = np.zeros(max(1, n), dtype=np.uint32)
        ln = np.zeros(max(1, n), dtype=np.uint32)
        self._ok(self.L.jsmpeg_hip_batch_ts_writes(self.h, stream, pts.ctypes.data, off.ctypes.data, ln.ctypes.data, n))
        return [(float(pts[i]), int(off[i]), int(ln[i])) for i in range(n)]

    def read_es(self, stream):
        n = self._ok(self.L.jsmpeg_hip_batch_read_es(self.h, stream, None, 0))
        out = np.empty(max(1, n), dtype=np.uint8)
        self._ok(self.L.jsmpeg_hip_batch_read_es(self.h, stream, out.ctypes.data, n))
        return out[:n]

    def upload_device(self, dev_ptr, total_bytes, begin, end, stream=None):
        begin = np.ascontiguousarray(begin, dtype=np.uint32)
        end = np.ascontiguousarray(end, dtype=np.uint32)
        self._ok(self.L.jsmpeg_hip_batch_upload_device(self.h, dev_ptr, total_bytes, len(begin), begin.ctypes.data,
                                                       end.ctypes.data, stream))

    def attach_device(self, dev_ptr, total_bytes, begin, end, stream=None):
        """upload_device without the copy: the next decode reads the caller's packed device buffer in place
        (16-byte aligned begins, 0xff between the ranges; include/jsmpeg_hip.h)"""
        begin = np.ascontiguousarray(begin, dtype=np.uint32)
        end = np.ascontiguousarray(end, dtype=np.uint32)
        self._ok(self.L.jsmpeg_hip_batch_attach_device(self.h, dev_ptr, total_bytes, len(begin), begin.ctypes.data,
                                                       end.ctypes.data, stream))

    def decode(self, stream=None, sync=True):
        n = self._ok(self.L.jsmpeg_hip_batch_decode(self.h, stream))
        if sync:
            self.sync()
        return n

    def sync(self):
        self._ok(self.L.jsmpeg_hip_batch_sync(self.h))

    @property
    def picture_count(self):
        return self.L.jsmpeg_hip_batch_picture_count(self.h)

    def picture_info(self, p):
        info = PictureInfo()
        self._ok(self.L.jsmpeg_hip_batch_picture_info(self.h, p, ctypes.byref(info)))
        return info

    def pictures(self):
        return [self.picture_info(p) for p in range(self.picture_count)]

    def read_frame(self, p):
        y = np.empty(self.luma_bytes, dtype=np.uint8)
        cr = np.empty(self.chroma_bytes, dtype=np.uint8)
        cb = np.empty(self.chroma_bytes, dtype=np.uint8)
        self._ok(self.L.jsmpeg_hip_batch_read_frame(self.h, p, y.ctypes.data, cr.ctypes.data, cb.ctypes.data))
        return y, cr, cb

    def read_frames(self, first, count):
        """pictures first .. first + count - 1 in one strided copy into pinned memory of the object's own (grown on demand):
        an array [count, luma_bytes + 2 * chroma_bytes], row k = Y | Cr | Cb of picture first + k; valid until the next call"""
        planes = self.luma_bytes + 2 * self.chroma_bytes
        if count <= 0:
            return np.empty((0, planes), dtype=np.uint8)
        if getattr(self, "_pin_bytes", 0) < count * planes:
            if getattr(self, "_pin", None):
                self.L.jsmpeg_hip_host_free(self._pin)
            self._pin_bytes = count * planes
            self._pin = self.L.jsmpeg_hip_host_alloc(self._pin_bytes)
            if not self._pin:
                self._pin_bytes = 0
                raise RuntimeError("jsmpeg_hip_host_alloc: " + last_error())
        self._ok(self.L.jsmpeg_hip_batch_read_frames(self.h, first, count, self._pin, planes))
        return np.ctypeslib.as_array((ctypes.c_uint8 * (count * planes)).from_address(self._pin)).reshape(count, planes)

    def frame_hashes(self):
        out = np.zeros(max(1, self.picture_count), dtype=np.uint64)
        self._ok(self.L.jsmpeg_hip_batch_frame_hashes(self.h, out.ctypes.data))
        return out[:self.picture_count]

    def render_rgba_device(self, first, count, dev_ptr, stream=None):
        """Renderer stage on the device: pictures [first, first + count) -> RGBA (width * height * 4 bytes each,
        display size) into the device buffer `dev_ptr`, enqueued on `stream`."""
        self._ok(self.L.jsmpeg_hip_batch_render_rgba(self.h, first, count, dev_ptr, stream))

    def read_rgba(self, p):
        """Picture p as RGBA uint8[height, width, 4]: device conversion, then a copy to the host."""
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._ok(self.L.jsmpeg_hip_batch_read_rgba(self.h, p, out.ctypes.data))
        return out

    def read_rgba_gl(self, p):
        """Picture p in the reference's WebGL renderer's arithmetic (bilinear chroma, float matrix): uint8[height, width, 4]."""
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        fn = self.L.jsmpeg_hip_batch_read_rgba_gl
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        self._ok(fn(self.h, p, out.ctypes.data))
        return out

    def timings(self):
        ms = (ctypes.c_float * 5)()
        self._ok(self.L.jsmpeg_hip_batch_timings(self.h, ms))
        return dict(index_ms=ms[0], host_ms=ms[1], parse_ms=ms[2], recon_ms=ms[3], total_ms=ms[4])

    def level_timings(self):
        """ms of every reconstruct launch of the last decode, in launch order ([0] = the intra level)"""
        ms = (ctypes.c_float * 64)()
        fn = self.L.jsmpeg_hip_batch_level_timings
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_uint32]
        n = self._ok(fn(self.h, ms, 64))
        return [float(ms[i]) for i in range(n)]

    def counters(self):
        c = (ctypes.c_uint64 * 8)()
        self._ok(self.L.jsmpeg_hip_batch_counters(self.h, c))
        return dict(start_codes=c[0], pictures=c[1], decoded=c[2], levels=c[3], slices=c[4], mb_per_picture=c[5],
                    uncovered_pictures=c[6], slice_codes=c[7])

    def link_streams(self, prev):
        """stream s continues stream prev[s] (< s) of the uploaded batch, -1: a stream of its own; None clears.  After upload / attach."""
        fn = self.L.jsmpeg_hip_batch_link_streams
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_uint32]
        if prev is None:
            self._ok(fn(self.h, None, 0))
        else:
            arr = (ctypes.c_int32 * len(prev))(*[int(x) for x in prev])
            self._ok(fn(self.h, arr, len(prev)))

    def seed_stream(self, stream, frame_last, frame_before_last):
        """device addresses (ints / None) of the frames of the decoded picture last / before last in front of `stream`"""
        fn = self.L.jsmpeg_hip_batch_seed_stream
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        self._ok(fn(self.h, stream, frame_last, frame_before_last))

    def uncovered(self):
        """per picture of the last decode: decoded and left macroblocks unwritten"""
        n = self.picture_count
        out = (ctypes.c_uint8 * max(1, n))()
        fn = self.L.jsmpeg_hip_batch_uncovered
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_uint32]
        k = self._ok(fn(self.h, out, n))
        return [int(out[i]) for i in range(k)]

    def set_reconstruct(self, plan):
        """0 / "levels": always one launch per dependency level; 1 / "auto": the engine's choice (include/jsmpeg_hip.h)"""
        fn = self.L.jsmpeg_hip_batch_set_reconstruct
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self._ok(fn(self.h, {"levels": 0, "auto": 1}.get(plan, plan)))

    def recon_info(self):
        """how the last decode reconstructed: launches (1 = the ordered launch), lockstep group, waits, status"""
        c = (ctypes.c_uint32 * 4)()
        fn = self.L.jsmpeg_hip_batch_recon_info
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        self._ok(fn(self.h, c))
        return dict(launches=c[0], group=c[1], waits=c[2], status=c[3])

    def stream_info(self, stream):
        """(has a sequence header, width, height, frame rate) of the stream's first sequence header as the last decode read it"""
        fn = self.L.jsmpeg_hip_batch_stream_info
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float)]
        w, h, r = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_float()
        has = self._ok(fn(self.h, stream, w, h, r))
        return bool(has), w.value, h.value, r.value

    @property
    def frame_pool_ptr(self):
        return self.L.jsmpeg_hip_batch_frame_pool(self.h)
