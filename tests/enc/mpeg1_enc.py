"""TEST INFRASTRUCTURE ONLY -- a small MPEG-1 video ENCODER, written independently of the synthetic stream generator
(jsmpeg_amd/csrc/synth_es.c), so that the parity fixtures are not all drawn from one author's idea of a stream.

synth_es.c draws random SYNTAX ELEMENTS: every macroblock its own uniform random vector, runs and levels from a fixed
mixture.  This file encodes PICTURES: procedural moving content -> block motion search (full-pel search + half-pel
refinement), float DCT, quantisation, and the decisions an encoder makes (skipped macroblocks, not-coded macroblocks,
intra fallback, quantiser changes) -- so the streams have the statistics of coded video: coherent vector fields, zero
vectors, long zero runs, sparse high frequencies, skipped runs, differential vectors that mostly repeat.

The encoder is open-loop (it predicts from the ORIGINAL previous picture, not from what a decoder would reconstruct):
its pictures drift from the source, which does not matter here -- the contract under test is "our decoder == the
reference decoder on a valid stream", not picture quality.  What it must get right is the SYNTAX (ISO 11172-2 as the
reference reads it) and the decoder-side predictor rules (DC and vector predictors, their
resets), restated here from the standard's clauses, with the VLC tables read from mpeg1_vlc_codes.h.

    python tests/enc/mpeg1_enc.py            # regenerates tests/golden/enc_*.m1v (container only: then make_golden_enc.py)
"""
import os
import sys

import numpy as np
from scipy.fft import dctn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jsmpeg_amd import spec_tables  # noqa: E402

T = spec_tables.load()
INV = {name: {v: b for b, v in T[name].items()} for name in ("MBA", "MBTYPE_I", "MBTYPE_P", "CBP", "MOTION", "DCSIZE_LUMA", "DCSIZE_CHROMA")}
COEFF = {v: b for b, v in T["DCT_COEFF"].items()}
ZZ = T["ZIGZAG"]
INTRA_Q = np.array(T["DEFAULT_INTRA_QUANT"], dtype=np.float64).reshape(8, 8)


class Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, nbits):
        self.acc = (self.acc << nbits) | (value & ((1 << nbits) - 1))
        self.n += nbits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def code(self, bits):
        self.put(int(bits, 2), len(bits))

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def start_code(self, c):
        self.align()
        self.out += bytes([0, 0, 1, c])


def content(width, height, n_frames, seed, pan=1.5, noise=1.5):
    """Procedural pictures: a panning textured background (pan = pixels per picture; 0: a static one, whose macroblocks an
    encoder skips), two moving textured rectangles, a little noise."""
    rng = np.random.RandomState(seed)
    big = 64
    yy, xx = np.mgrid[0:height + 2 * big, 0:width + 2 * big].astype(np.float64)
    tex = 110 + 50 * np.sin(xx / 17.0) * np.cos(yy / 23.0) + 30 * np.sin((xx + 2 * yy) / 7.0) + 12 * rng.randn(*xx.shape).cumsum(axis=1) / 8
    tex_u = 128 + 40 * np.sin(xx / 31.0 + 1) + 20 * np.cos(yy / 19.0)
    tex_v = 128 + 40 * np.cos(xx / 27.0) - 25 * np.sin(yy / 13.0 + 2)
    frames = []
    for t in range(n_frames):
        ox, oy = big + int(round(pan * t)), big + int(round(pan / 3.0 * t * ((t // 4) % 2 * 2 - 1)))
        y = tex[oy:oy + height, ox:ox + width].copy()
        u = tex_u[oy:oy + height:2, ox:ox + width:2].copy()
        v = tex_v[oy:oy + height:2, ox:ox + width:2].copy()
        for k, (w0, h0, vx, vy, lum) in enumerate(((48, 40, 3, 1, 200), (36, 60, -2, 2, 40))):
            x0 = int((30 + 90 * k + vx * t) % max(1, width - w0)) & ~1
            y0 = int((20 + 50 * k + vy * t) % max(1, height - h0)) & ~1
            y[y0:y0 + h0, x0:x0 + w0] = lum + 25 * np.sin(np.arange(w0) / 3.0)[None, :] * np.cos(np.arange(h0) / 4.0)[:, None]
            u[y0 // 2:(y0 + h0) // 2, x0 // 2:(x0 + w0) // 2] = 90 + 60 * k
            v[y0 // 2:(y0 + h0) // 2, x0 // 2:(x0 + w0) // 2] = 170 - 70 * k
        y += noise * rng.randn(*y.shape)
        frames.append((np.clip(y, 0, 255), np.clip(u, 0, 255), np.clip(v, 0, 255)))
    return frames


def pad_planes(frame, cw, ch):
    y, u, v = frame
    Y = np.pad(y, ((0, ch - y.shape[0]), (0, cw - y.shape[1])), mode="edge")
    U = np.pad(u, ((0, ch // 2 - u.shape[0]), (0, cw // 2 - u.shape[1])), mode="edge")
    V = np.pad(v, ((0, ch // 2 - v.shape[0]), (0, cw // 2 - v.shape[1])), mode="edge")
    return Y, U, V


def predict(plane, x, y, mvh, mvv, n):
    """n x n prediction at (x, y) displaced by (mvh, mvv) half-pels: the four half-pel cases of ISO 11172-2 2.4.4.2."""
    H, V, oh, ov = mvh >> 1, mvv >> 1, mvh & 1, mvv & 1
    a = plane[y + V:y + V + n + ov, x + H:x + H + n + oh]
    if oh:
        a = (a[:, :-1] + a[:, 1:]) / 2.0
    if ov:
        a = (a[:-1, :] + a[1:, :]) / 2.0
    return a


def mv_ok(cw, ch, col, row, mvh, mvv):
    """every pixel a decoder reads for this vector lies inside the coded picture (luma and chroma)"""
    H, V, oh, ov = mvh >> 1, mvv >> 1, mvh & 1, mvv & 1
    x0, y0 = col * 16 + H, row * 16 + V
    if x0 < 0 or y0 < 0 or x0 + 15 + oh > cw - 1 or y0 + 15 + ov > ch - 1:
        return False
    c_h, c_v = int(mvh / 2), int(mvv / 2)        # toward zero, like the decoder
    cH, cV, coh, cov = c_h >> 1, c_v >> 1, c_h & 1, c_v & 1
    cx0, cy0 = col * 8 + cH, row * 8 + cV
    return not (cx0 < 0 or cy0 < 0 or cx0 + 7 + coh > cw // 2 - 1 or cy0 + 7 + cov > ch // 2 - 1)


def put_coeffs(w, levels_zz, first_is_special):
    """run/level pairs of the non-zero entries of `levels_zz` (scan order), then end_of_block"""
    run, first = 0, first_is_special
    for lv in levels_zz:
        if lv == 0:
            run += 1
            continue
        mag = abs(lv)
        if run == 0 and mag == 1:
            w.code("1" if first else "11")
            w.put(lv < 0, 1)
        elif (run, mag) in COEFF:
            w.code(COEFF[(run, mag)])
            w.put(lv < 0, 1)
        else:
            w.code(T["DCT_ESCAPE"])
            w.put(run, 6)
            if -127 <= lv <= 127:
                w.put(lv & 0xFF, 8)
            elif lv > 0:
                w.put(0x00, 8); w.put(lv, 8)
            else:
                w.put(0x80, 8); w.put(lv + 256, 8)
        run, first = 0, False
    w.code("10")


def scan(block):
    return [int(block.flat[ZZ[i]]) for i in range(64)]


def put_motion(w, d, r_size):
    f = 1 << r_size
    if d == 0:
        w.code(INV["MOTION"][0]); return
    if f == 1:
        w.code(INV["MOTION"][d]); return
    ad = abs(d) - 1
    mag = (ad >> r_size) + 1
    w.code(INV["MOTION"][-mag if d < 0 else mag])
    w.put(ad & (f - 1), r_size)


def encode(width, height, n_frames, gop=6, qscale=6, f_code=1, seed=1, half_pel=True, quant_changes=True, pan=1.5, noise=1.5):
    mbw, mbh = (width + 15) >> 4, (height + 15) >> 4
    cw, ch = mbw * 16, mbh * 16
    frames = [pad_planes(f, cw, ch) for f in content(width, height, n_frames, seed, pan, noise)]
    rng = np.random.RandomState(seed + 1000)
    w = Bits()
    r_size = f_code - 1
    rng_mv = 16 << r_size                      # vectors in [-rng_mv, rng_mv - 1] half-pels
    search = (rng_mv // 2) - 1                 # full-pel search radius so that a half-pel refinement stays in range
    pic_offsets = []
    for t in range(n_frames):
        ptype = 1 if t % gop == 0 else 2
        if ptype == 1:
            w.start_code(0xB3)
            w.put(width, 12); w.put(height, 12); w.put(1, 4); w.put(5, 4); w.put(0x3FFFF, 18); w.put(1, 1); w.put(20, 10); w.put(0, 1)
            w.put(0, 1); w.put(0, 1)
            w.start_code(0xB8)
            sec, pic = t // 30, t % 30
            w.put(0, 1); w.put(0, 5); w.put(0, 6); w.put(1, 1); w.put(sec % 60, 6); w.put(pic, 6); w.put(1, 1); w.put(0, 1)
        w.align()
        pic_offsets.append(len(w.out))
        w.start_code(0x00)
        w.put(t % gop, 10); w.put(ptype, 3); w.put(0xFFFF, 16)
        if ptype == 2:
            w.put(0, 1); w.put(f_code, 3)
        w.put(0, 1)                             # extra_bit_picture
        Y, U, V = frames[t]
        if ptype == 2:
            PY, PU, PV = frames[t - 1]
        for row in range(mbh):
            w.start_code(row + 1)
            q = qscale
            w.put(q, 5); w.put(0, 1)
            dc_pred = [128.0, 128.0, 128.0]
            pmh = pmv = 0
            last_coded = -1                     # address (in the row) of the previous coded macroblock
            for col in range(mbw):
                y0, x0 = row * 16, col * 16
                cur = [Y[y0:y0 + 8, x0:x0 + 8], Y[y0:y0 + 8, x0 + 8:x0 + 16], Y[y0 + 8:y0 + 16, x0:x0 + 8], Y[y0 + 8:y0 + 16, x0 + 8:x0 + 16],
                       U[y0 // 2:y0 // 2 + 8, x0 // 2:x0 // 2 + 8], V[y0 // 2:y0 // 2 + 8, x0 // 2:x0 // 2 + 8]]
                new_q = q
                if quant_changes and rng.randint(0, 14) == 0:
                    new_q = int(np.clip(q + rng.randint(-2, 3), 2, 14))
                intra = ptype == 1
                mvh = mvv = 0
                if ptype == 2:
                    # ---- motion search on luma: full-pel, then the eight half-pel neighbours ----
                    blk = Y[y0:y0 + 16, x0:x0 + 16]
                    best = (np.abs(blk - PY[y0:y0 + 16, x0:x0 + 16]).sum() - 64, 0, 0)       # a little bias toward the zero vector
                    for dy in range(-search, search + 1):
                        for dx in range(-search, search + 1):
                            if not mv_ok(cw, ch, col, row, 2 * dx, 2 * dy):
                                continue
                            sad = np.abs(blk - PY[y0 + dy:y0 + dy + 16, x0 + dx:x0 + dx + 16]).sum()
                            if sad < best[0]:
                                best = (sad, 2 * dx, 2 * dy)
                    if half_pel:
                        bh, bv = best[1], best[2]
                        for hv in range(-1, 2):
                            for hh in range(-1, 2):
                                mh, mv = bh + hh, bv + hv
                                if (hh or hv) and -rng_mv <= mh < rng_mv and -rng_mv <= mv < rng_mv and mv_ok(cw, ch, col, row, mh, mv):
                                    sad = np.abs(blk - predict(PY, x0, y0, mh, mv, 16)).sum()
                                    if sad < best[0]:
                                        best = (sad, mh, mv)
                    intra_cost = np.abs(blk - blk.mean()).sum()
                    if intra_cost + 500 < best[0]:
                        intra = True
                    else:
                        mvh, mvv = best[1], best[2]
                # ---- quantised levels ----
                levels, cbp = [], 0
                if intra:
                    for b in range(6):
                        c = dctn(cur[b], norm="ortho")
                        lv = np.rint(c * 8.0 / (new_q * INTRA_Q))
                        lv = np.clip(lv, -255, 255).astype(np.int64)
                        lv.flat[0] = int(np.clip(np.rint(c[0, 0] / 8.0), 0, 255))
                        levels.append(lv)
                    cbp = 0x3F
                else:
                    c_h, c_v = int(mvh / 2), int(mvv / 2)
                    preds = [predict(PY, x0, y0, mvh, mvv, 16)[:8, :8], predict(PY, x0, y0, mvh, mvv, 16)[:8, 8:], predict(PY, x0, y0, mvh, mvv, 16)[8:, :8],
                             predict(PY, x0, y0, mvh, mvv, 16)[8:, 8:], predict(PU, x0 // 2, y0 // 2, c_h, c_v, 8), predict(PV, x0 // 2, y0 // 2, c_h, c_v, 8)]
                    for b in range(6):
                        c = dctn(cur[b] - preds[b], norm="ortho")
                        lv = np.fix(c * 8.0 / (new_q * 16.0))
                        lv = np.clip(lv, -255, 255).astype(np.int64)
                        levels.append(lv)
                        if np.any(lv):
                            cbp |= 0x20 >> b
                # ---- macroblock type decisions ----
                first_or_last = col == 0 or col == mbw - 1
                has_mv = (mvh != 0 or mvv != 0)
                if not intra and cbp == 0 and not has_mv and not first_or_last:
                    # skipped: nothing is written; a decoder copies the co-located macroblock and resets its predictors
                    dc_pred = [128.0, 128.0, 128.0]
                    pmh = pmv = 0
                    continue
                if not intra and cbp == 0 and not has_mv:
                    has_mv = True              # first / last macroblock of a slice must be coded: "MC, not coded" with a zero vector
                inc = col - last_coded
                if last_coded < 0:
                    inc = col + 1               # the first increment of a slice is relative to the row start
                while inc > 33:
                    w.code(INV["MBA"][35]); inc -= 33
                w.code(INV["MBA"][inc])
                if last_coded >= 0 and col - last_coded > 1:
                    dc_pred = [128.0, 128.0, 128.0]
                    pmh = pmv = 0
                last_coded = col
                use_q = new_q != q and (intra or cbp != 0)
                if ptype == 1:
                    w.code(INV["MBTYPE_I"][0x11 if use_q else 0x01])
                else:
                    if intra:
                        mtype = 0x11 if use_q else 0x01
                    elif cbp and has_mv:
                        mtype = 0x1A if use_q else 0x0A
                    elif cbp:
                        mtype = 0x12 if use_q else 0x02
                    else:
                        mtype = 0x08
                    w.code(INV["MBTYPE_P"][mtype])
                if use_q:
                    w.put(new_q, 5)
                    q = new_q
                elif new_q != q:
                    # the quantiser did not travel: requantise is not worth it, the levels were made with new_q -- any levels are valid
                    pass
                if intra:
                    pmh = pmv = 0
                    for b in range(6):
                        comp = 0 if b < 4 else (1 if b == 4 else 2)
                        dc = int(levels[b].flat[0])
                        diff = dc - int(dc_pred[comp])
                        # keep the decoder's predictor inside 0..255 whatever the history
                        size = 0 if diff == 0 else int(abs(diff)).bit_length()
                        w.code(INV["DCSIZE_LUMA" if b < 4 else "DCSIZE_CHROMA"][size])
                        if size:
                            w.put(diff if diff > 0 else diff + (1 << size) - 1, size)
                        dc_pred[comp] = dc
                        zz = scan(levels[b])
                        put_coeffs(w, zz[1:], False)
                else:
                    dc_pred = [128.0, 128.0, 128.0]
                    if has_mv and mtype != 0x02 and mtype != 0x12:
                        for cur_mv, prev in ((mvh, pmh), (mvv, pmv)):
                            d = cur_mv - prev
                            if d < -rng_mv:
                                d += 2 * rng_mv
                            elif d >= rng_mv:
                                d -= 2 * rng_mv
                            put_motion(w, d, r_size)
                        pmh, pmv = mvh, mvv
                    else:
                        pmh = pmv = 0           # a P macroblock without a vector zeroes the decoder's predictor
                    if cbp:
                        w.code(INV["CBP"][cbp])
                        for b in range(6):
                            if cbp & (0x20 >> b):
                                put_coeffs(w, scan(levels[b]), True)
        w.align()
    w.start_code(0xB7)
    pic_offsets.append(len(w.out) - 4)
    pic_offsets[0] = 0                                  # the first picture's range begins with the sequence header
    es = np.frombuffer(bytes(w.out), dtype=np.uint8).copy()
    # no start code may appear inside a slice: count them (per picture: the picture code + one per row; per GOP two headers; the end code)
    n_codes = int(np.count_nonzero((es[:-3] == 0) & (es[1:-2] == 0) & (es[2:-1] == 1)))
    n_gops = (n_frames + gop - 1) // gop
    assert n_codes == n_frames * (1 + mbh) + 2 * n_gops + 1, "start code emulated inside a slice (%d codes)" % n_codes
    return es, np.array(pic_offsets, dtype=np.uint32)


CASES = {
    # name: encode() arguments
    "enc_pan_176x144": dict(width=176, height=144, n_frames=13, gop=6, qscale=6, f_code=1, seed=1),
    "enc_wide_search_208x160": dict(width=208, height=160, n_frames=10, gop=5, qscale=4, f_code=2, seed=2),
    "enc_coarse_fullpel_160x128": dict(width=160, height=128, n_frames=12, gop=12, qscale=12, f_code=1, seed=3, half_pel=False),
    # full size, a STATIC background behind the moving rectangles: rows of skipped macroblocks (increments beyond 33:
    # macroblock_escape), zero vectors, the 2-D tiles and the 16-wide remainder column of a 1080p plane with coded video in them
    "enc_static_1920x1080": dict(width=1920, height=1080, n_frames=3, gop=3, qscale=8, f_code=1, seed=4, pan=0.0, noise=0.0),
}

if __name__ == "__main__":
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, kw in CASES.items():
        es, offs = encode(**kw)
        assert not np.any((es[:-3] == 0) & (es[1:-2] == 0) & (es[2:-1] == 1) & (es[3:] > 0xAF) & (es[3:] < 0xB3)), "reserved start code emulated"
        es.tofile(os.path.join(out_dir, name + ".m1v"))
        np.save(os.path.join(out_dir, name + ".offsets.npy"), offs)
        print("%-30s %d pictures, %d bytes" % (name, len(offs) - 1, len(es)))
