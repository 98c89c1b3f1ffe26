"""TEST INFRASTRUCTURE ONLY -- ctypes helpers over oracle/libmpeg1_oracle.so for the stages either side of the decode
path: the reference's Canvas2D colour conversion (oracle/ycbcr_oracle.c, reference src/canvas2d.js:53-122) and its TS
demuxer (oracle/ts_oracle.c, reference src/ts.js:25-210).  Imported by tests/, tools/, tests/golden/make_golden_*.py
and __graft_entry__.smoke() as the thing to compare against; never by the product package."""
import ctypes

import numpy as np


def oracle_rgba(oracle_path, y, cr, cb, width, height):
    """CHECKER ONLY: the reference's Canvas2D colour conversion restated on the CPU (oracle/ycbcr_oracle.c);
    uint8[height, width, 4]."""
    lib = ctypes.CDLL(oracle_path)
    fn = lib.ycbcr_oracle_to_rgba
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    y, cr, cb = (np.ascontiguousarray(a, dtype=np.uint8) for a in (y, cr, cb))
    out = np.empty((height, width, 4), dtype=np.uint8)
    fn(y.ctypes.data, cr.ctypes.data, cb.ctypes.data, width, height, out.ctypes.data)
    return out


def oracle_rgba_gl(oracle_path, y, cr, cb, width, height):
    """CHECKER ONLY: the reference's WebGL colour conversion (bilinear chroma, float BT.601 matrix) restated in float64
    (oracle/ycbcr_oracle.c; unpinned: no WebGL runs here); uint8[height, width, 4]."""
    lib = ctypes.CDLL(oracle_path)
    fn = lib.ycbcr_oracle_to_rgba_gl
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    y, cr, cb = (np.ascontiguousarray(a, dtype=np.uint8) for a in (y, cr, cb))
    out = np.empty((height, width, 4), dtype=np.uint8)
    fn(y.ctypes.data, cr.ctypes.data, cb.ctypes.data, width, height, out.ctypes.data)
    return out


class _TsWrite(ctypes.Structure):
    _fields_ = [("pts", ctypes.c_double), ("offset", ctypes.c_uint32), ("length", ctypes.c_uint32)]


def oracle_ts_demux(oracle_path, ts, stream_id=0xE0, write_sizes=None):
    """CHECKER ONLY: the reference's TS demuxer restated on the CPU (oracle/ts_oracle.c): one write() of the whole
    buffer, or the buffer in write() calls of `write_sizes` bytes.  Returns (es bytes, [(pts, offset, length)] per
    destination.write call)."""
    lib = ctypes.CDLL(oracle_path)
    ts = np.ascontiguousarray(ts, dtype=np.uint8)
    es = np.zeros(len(ts) + 16, dtype=np.uint8)
    cap = len(ts) // 94 + 16
    writes = (_TsWrite * cap)()
    n_es = ctypes.c_size_t()
    if write_sizes is None:
        fn = lib.ts_oracle_demux
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                       ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p, ctypes.c_int]
        n = fn(ts.ctypes.data, len(ts), stream_id, es.ctypes.data, len(es), ctypes.byref(n_es), writes, cap)
    else:
        fn = lib.ts_oracle_demux_writes
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                       ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p, ctypes.c_int]
        ws = (ctypes.c_uint64 * len(write_sizes))(*[int(x) for x in write_sizes])
        n = fn(ts.ctypes.data, len(ts), ws, len(write_sizes), stream_id, es.ctypes.data, len(es), ctypes.byref(n_es), writes, cap)
    if n < 0 or n > cap:
        raise RuntimeError("ts_oracle_demux failed (%d)" % n)
    return es[:n_es.value].copy(), [(writes[i].pts, writes[i].offset, writes[i].length) for i in range(n)]
