"""Renderer stage: Y/Cr/Cb -> RGBA, the reference's Canvas2D integer conversion
(src/canvas2d.js:53-122).  The fixtures tests/golden/rgba_*.json were agreed between the unmodified reference
(mpeg1.js + canvas2d.js under Node) and the CPU restatement (tests/golden/make_golden_rgba.py).
CPU tests pin the restatement to the fixtures; GPU tests compare k_rgba (through the C ABI) with both."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from jsmpeg_amd import cabi, synth
from oracle import checkers

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "rgba_*.json")))
IDS = [os.path.basename(p)[5:-5] for p in FIXTURES]


def load_case(path):
    fx = json.load(open(path))
    es, offs = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
    assert hashlib.md5(es.tobytes()).hexdigest() == fx["es_md5"]
    return fx, es, offs


def md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_oracle_matches_reference_fixture(path, libs):
    fx, es, _ = load_case(path)
    frames, _, info = cabi.decode_stream(libs["oracle"], es, keep="planes")
    got = [md5(checkers.oracle_rgba(libs["oracle"], y, cr, cb, info["width"], info["height"])) for y, cr, cb in frames]
    assert got == fx["rgba_md5"]


def test_oracle_odd_width_is_sheared_like_the_reference(libs):
    """With an odd width the reference's running indices drift by one pixel per row pair (canvas2d.js:74-76, 115-119:
    the inner loop advances 2 * cols pixels, the row step adds `width`), and what the loop never writes keeps
    resize()'s 255 fill (canvas2d.js:33).  The fixtures pin that against the reference; this spells it out."""
    w, h, cw, ch = 17, 33, 32, 48
    rng = np.random.default_rng(1)
    y = rng.integers(0, 256, cw * ch, dtype=np.uint8)
    cr = rng.integers(0, 256, cw * ch // 4, dtype=np.uint8)
    cb = rng.integers(0, 256, cw * ch // 4, dtype=np.uint8)
    out = checkers.oracle_rgba(libs["oracle"], y, cr, cb, w, h).reshape(-1, 4)
    cols, rows = w >> 1, h >> 1
    S = 2 * cols + w
    assert (out[:, 3] == 255).all()
    assert (out[rows * S:] == 255).all()                       # never reached
    assert all((out[rp * S + 2 * cols] == 255).all() for rp in range(rows))   # the one-pixel gap between the two lines
    # first pixel of row pair 3, line 2: luma index 3 * (2 * cols + 2 * cw - w) + cw, chroma index 3 * (cw / 2)
    yy, c_r, c_b = int(y[3 * (2 * cols + 2 * cw - w) + cw]), int(cr[3 * (cw >> 1)]), int(cb[3 * (cw >> 1)])
    r = (c_r + ((c_r * 103) >> 8)) - 179
    g = ((c_b * 88) >> 8) - 44 + ((c_r * 183) >> 8) - 91
    b = (c_b + ((c_b * 198) >> 8)) - 227
    want = [min(max(yy + r, 0), 255), min(max(yy - g, 0), 255), min(max(yy + b, 0), 255), 255]
    assert out[3 * S + w].tolist() == want


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_decoder_abi_render_rgba(path, hip_lib, libs):
    """One-picture interface: decode, then jsmpeg_hip_decoder_render_rgba (device conversion, RGBA to the host)."""
    fx, es, _ = load_case(path)
    got = []
    with cabi.Mpeg1Decoder(hip_lib, len(es) + 1024, cabi.MODE_EXPAND) as dec:
        dec.write(es)
        while dec.decode():
            rgba = dec.render_rgba()
            if len(got) < 2:   # also against the restatement on the very planes this decoder returned
                y, cr, cb = dec.planes()
                assert np.array_equal(rgba, checkers.oracle_rgba(libs["oracle"], y, cr, cb, dec.width, dec.height))
            got.append(md5(rgba))
    assert got == fx["rgba_md5"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_batch_render_rgba(path, hip_lib):
    """Batch interface: pictures of the pool -> RGBA (jsmpeg_hip_batch_render_rgba behind read_rgba)."""
    from jsmpeg_amd import batch as jb
    fx, es, _ = load_case(path)
    n, w, h = fx["n_frames"], fx["width"], fx["height"]
    with jb.Batch(w, h, 1, n + 2, len(es) + 1024) as b:
        b.upload([es])
        assert b.decode() == n
        assert [md5(b.read_rgba(p)) for p in range(n)] == fx["rgba_md5"]
        with pytest.raises(RuntimeError):
            b.read_rgba(n)


# ---- the reference's other renderer form: WebGL (src/webgl.js:259-281).  No WebGL implementation runs in the build
# container and a browser's result depends on its GPU (shader precision, filter hardware): the restatement is float64 and
# UNPINNED; what is checked is its arithmetic against a plain numpy statement of the shader (CPU) and the device kernel
# against it within 1 LSB (GPU) ----

def _numpy_webgl(y, cr, cb, width, height):
    cw, h2 = ((width + 15) >> 4) << 4, height >> 1
    cw2 = cw >> 1
    Y = y.reshape(-1, cw)[:height, :width].astype(np.float64) / 255.0
    px, py = np.meshgrid(np.arange(width), np.arange(height))
    u, v = (px + 0.5) / cw * cw2 - 0.5, (py + 0.5) / height * h2 - 0.5
    fu, fv = np.floor(u), np.floor(v)
    ax, ay = u - fu, v - fv
    x0, x1 = np.clip(fu, 0, cw2 - 1).astype(int), np.clip(fu + 1, 0, cw2 - 1).astype(int)
    y0, y1 = np.clip(fv, 0, h2 - 1).astype(int), np.clip(fv + 1, 0, h2 - 1).astype(int)

    def sample(P):
        P = P.reshape(-1, cw2).astype(np.float64)
        return ((1 - ay) * ((1 - ax) * P[y0, x0] + ax * P[y0, x1]) + ay * ((1 - ax) * P[y1, x0] + ax * P[y1, x1])) / 255.0
    CR, CB = sample(cr), sample(cb)
    M = np.array([[1.16438, 0.0, 1.59603, -0.87079], [1.16438, -0.39176, -0.81297, 0.52959], [1.16438, 2.01723, 0.0, -1.08139]])
    # gl_FragColor = vec4(y, cr, cb, 1) * rec601, the shader's cr = true Cb, its cb = true Cr
    vec = np.stack([Y, CB, CR, np.ones_like(Y)], axis=-1)
    rgb = np.clip(vec @ M.T, 0.0, 1.0)
    out = np.full((height, width, 4), 255, np.uint8)
    out[..., :3] = np.floor(rgb * 255.0 + 0.5).astype(np.uint8)
    return out


def test_webgl_restatement_is_the_shader(libs):
    rng = np.random.default_rng(5)
    for w, h in ((32, 16), (17, 33), (352, 288), (30, 2)):
        cw, ch = ((w + 15) >> 4) << 4, ((h + 15) >> 4) << 4
        y = rng.integers(0, 256, cw * ch, dtype=np.uint8)
        cr = rng.integers(0, 256, cw * ch // 4, dtype=np.uint8)
        cb = rng.integers(0, 256, cw * ch // 4, dtype=np.uint8)
        got = checkers.oracle_rgba_gl(libs["oracle"], y, cr, cb, w, h)
        want = _numpy_webgl(y, cr, cb, w, h)
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and (got == want).mean() > 0.999
    # a grey ramp with chroma 128: nearly grey (the shader's offsets are exact for 128 / 255 only to 0.006), black at 16, white at 235
    y = np.tile(np.arange(256, dtype=np.uint8), 16)
    neutral = np.full(256 * 16 // 4, 128, np.uint8)
    out = checkers.oracle_rgba_gl(libs["oracle"], y, neutral, neutral, 256, 16)
    assert (np.abs(out[..., 0].astype(int) - out[..., 1]) <= 2).all() and out[0, 16, 0] <= 1 and out[0, 235, 0] >= 254


@pytest.mark.gpu
def test_batch_render_rgba_webgl_form(hip_lib, libs):
    """k_rgba_gl through the C ABI against the float64 restatement: at most 1 LSB apart, almost everywhere equal."""
    from jsmpeg_amd import batch as jb
    for w, h, n in ((352, 288, 4), (17, 33, 3), (1280, 96, 2)):
        es, _ = synth.generate_config("cfg1_720p", n_frames=n, width=w, height=h)
        frames, _, info = cabi.decode_stream(libs["oracle"], es, keep="planes")
        with jb.Batch(w, h, 1, n + 2, len(es) + 1024) as b:
            b.upload([es])
            assert b.decode() == n
            for p, (y, cr, cb) in enumerate(frames):
                got = b.read_rgba_gl(p)
                want = checkers.oracle_rgba_gl(libs["oracle"], y, cr, cb, w, h)
                d = np.abs(got.astype(int) - want.astype(int))
                assert d.max() <= 1 and (d == 0).mean() > 0.99, (w, h, p, d.max(), (d == 0).mean())
