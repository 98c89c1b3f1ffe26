"""The Node.js host side: N-API addon + JSMpeg.Decoder.MPEG1VideoHIP."""
import glob
import hashlib
import json
import os
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT, have_reference
from jsmpeg_amd import build, synth

NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node not installed")


def _ts_for(case):
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "frames_%s.json" % case)))
    es, offs = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
    assert hashlib.md5(es.tobytes()).hexdigest() == fx["es_md5"]
    f = tempfile.NamedTemporaryFile(suffix=".ts", delete=False)
    f.write(synth.mux_ts(es, offs).tobytes())
    f.close()
    return fx, f.name


def test_addon_loads_and_exports_the_abi():
    addon = build.build_addon()
    assert addon and os.path.exists(addon)
    out = subprocess.check_output([NODE, "-e", "const a=require(%r);console.log(JSON.stringify(Object.keys(a)))" % addon])
    names = set(json.loads(out))
    assert {"create", "destroy", "bufferWrite", "getIndex", "setIndex", "hasSequenceHeader", "getFrameRate",
            "getCodedSize", "getWidth", "getHeight", "decode", "getPlanes", "renderRGBA", "deviceCount",
            "lastError", "batchCreate", "batchDestroy", "batchUpload", "batchUploadTS", "batchDecode", "batchPictureInfo",
            "batchTsWrites", "batchReadPlanes", "batchReadRGBA", "batchGeometry", "batchTimings"} <= names


def test_class_fails_loudly_without_gpu():
    """No GPU in the build container: constructing is fine (lazy), the first write must throw -- never a silent
    JS/CPU fallback."""
    from conftest import have_gpu
    if have_gpu():
        pytest.skip("a GPU is present")
    build.build_addon()
    script = ("const {install}=require(%r);const {MPEG1VideoHIP}=install();const d=new MPEG1VideoHIP({});"
              "try{d.write(0,[new Uint8Array(8)]);console.log('NO THROW')}catch(e){console.log('THROWS:'+e.message)}"
              % os.path.join(ROOT, "jsmpeg_amd", "js", "mpeg1-hip.js"))
    out = subprocess.check_output([NODE, "-e", script]).decode()
    assert out.startswith("THROWS:") and "no CPU fallback" in out


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference")
@pytest.mark.parametrize("mode", ["static", "streaming"])
@pytest.mark.parametrize("case", ["custom_quant_escapes", "cfg0_240p_intra"])
def test_class_is_a_dropin_for_the_wasm_wrapper(case, mode):
    """Same resize/render/onVideoDecode/currentTime/index/seek event sequence as the reference's
    JSMpeg.Decoder.MPEG1VideoWASM, driven by the reference's own TS demuxer (binding = the reference's wasm
    exports, so only the class is under test)."""
    fx, ts = _ts_for(case)
    try:
        args = [NODE, os.path.join(ROOT, "tests", "js", "class_vs_reference.js"), ts]
        out = json.loads(subprocess.check_output(args + (["streaming"] if mode == "streaming" else [])))
    finally:
        os.unlink(ts)
    assert out["same"], out
    assert out["renders"] >= fx["n_frames"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["static", "streaming"])
@pytest.mark.parametrize("case", ["cfg0_240p_intra", "cfg1_720p", "custom_quant_escapes", "odd_size_17x33"])
def test_node_class_on_gpu_matches_golden(case, mode, hip_lib):
    """TS file -> ts-demux.js -> MPEG1VideoHIP -> real addon -> HIP kernels; rendered planes vs golden."""
    build.build_addon()
    fx, ts = _ts_for(case)
    try:
        args = [NODE, os.path.join(ROOT, "tests", "js", "hip_decode_ts.js"), ts]
        out = json.loads(subprocess.check_output(args + (["streaming"] if mode == "streaming" else [])))
    finally:
        os.unlink(ts)
    assert out["hashes"] == fx["frame_md5"]
    assert out["sizes"] == [[fx["info"]["width"], fx["info"]["height"]]]
    assert abs(out["frameRate"] - 30.0) < 1e-6


def test_renderer_refuses_to_convert_without_the_device_path():
    """JSMpeg.Renderer.HIPRGBA is the device path: without a live MPEG1VideoHIP handle render() throws, it never
    converts in JS."""
    script = ("const {install}=require(%r);const {HIPRGBA}=install();const r=new HIPRGBA({});r.resize(16,16);"
              "try{r.render(new Uint8Array(256),new Uint8Array(64),new Uint8Array(64));console.log('NO THROW')}"
              "catch(e){console.log('THROWS:'+e.message)}" % os.path.join(ROOT, "jsmpeg_amd", "js", "renderer-hip.js"))
    out = subprocess.check_output([NODE, "-e", script]).decode()
    assert out.startswith("THROWS:")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cfg0_240p_intra", "cif_352x288", "odd_size_17x33"])
def test_node_renderer_on_gpu_matches_reference_canvas2d(case, hip_lib):
    """TS file -> MPEG1VideoHIP -> Renderer.HIPRGBA: imageData.data per picture equals what the reference's
    Canvas2D renderer produced for the same stream (tests/golden/rgba_*.json)."""
    build.build_addon()
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "rgba_%s.json" % case)))
    es, offs = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
    assert hashlib.md5(es.tobytes()).hexdigest() == fx["es_md5"]
    f = tempfile.NamedTemporaryFile(suffix=".ts", delete=False)
    f.write(synth.mux_ts(es, offs).tobytes())
    f.close()
    try:
        out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "hip_render_rgba.js"), f.name]))
    finally:
        os.unlink(f.name)
    assert out["hashes"] == fx["rgba_md5"]
    assert (out["width"], out["height"]) == (fx["width"], fx["height"])


def test_batch_class_fails_loudly_without_gpu():
    from conftest import have_gpu
    if have_gpu():
        pytest.skip("a GPU is present")
    build.build_addon()
    script = ("const {install}=require(%r);const {HIPBatch}=install();"
              "try{new HIPBatch({width:320,height:240});console.log('NO THROW')}catch(e){console.log('THROWS:'+e.message)}"
              % os.path.join(ROOT, "jsmpeg_amd", "js", "batch-hip.js"))
    out = subprocess.check_output([NODE, "-e", script]).decode()
    assert out.startswith("THROWS:") and "no CPU fallback" in out


@pytest.mark.gpu
def test_node_batch_ts_in_frames_out(hip_lib):
    """Three TS files -> JSMpeg.HIPBatch (device demux, batch decode, device RGBA) under Node: planes against the
    frame fixture, RGBA against the reference-Canvas2D fixture, pts as ts.js reports them."""
    build.build_addon()
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "rgba_cif_352x288.json")))
    paths, want = [], []
    try:
        for s in range(3):
            es, offs = synth.generate_config(fx["config"], n_frames=fx["n_frames"], stream=s, **fx["overrides"])
            f = tempfile.NamedTemporaryFile(suffix=".ts", delete=False)
            f.write(synth.mux_ts(es, offs).tobytes())
            f.close()
            paths.append(f.name)
            want.append(cabi_md5_frames(es))
        out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "hip_batch_ts.js"), "352", "288"] + paths))
    finally:
        for p in paths:
            os.unlink(p)
    assert out["pictures"] == 3 * fx["n_frames"]
    for s in range(3):
        assert out["streams"][s]["planes"] == want[s]
        pts = out["streams"][s]["pts"]
        assert len(pts) == fx["n_frames"] and abs(pts[0] - 0.1) < 1e-9 and abs(pts[1] - pts[0] - 1 / 30) < 1e-4
    assert out["streams"][0]["rgba"] == fx["rgba_md5"]          # stream 0 is the fixture's stream


def cabi_md5_frames(es):
    """md5(Y|Cr|Cb) per frame from the oracle (checker)."""
    from jsmpeg_amd import cabi
    return cabi.decode_stream(build.build_oracle(), es)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["static", "streaming"])
def test_player_hip_over_the_real_addon(mode, hip_lib):
    """JSMpeg.PlayerHIP's decoder selection (player-hip.js; reference src/player.js:35-52) with the REAL addon: an A/V
    transport stream -> Player (tests/js/mini_player.js stands in for the reference's player.js, which cannot travel
    to the GPU box; the reference's own Player is covered in the container, tests/test_player_hip.py) -> the HIP
    classes must be the ones constructed, the names restored, and every picture / audio frame must match the fixtures."""
    import numpy as np
    from test_mp2_gpu import _av_ts
    from jsmpeg_amd import cabi
    build.build_addon()
    ts, es, afx, _ = _av_ts(13, "stereo_44k_192", 3)
    f = tempfile.NamedTemporaryFile(suffix=".ts", delete=False)
    f.write(ts.tobytes())
    f.close()
    try:
        args = [NODE, os.path.join(ROOT, "tests", "js", "player_hip_gpu.js"), f.name] + (["streaming"] if mode == "streaming" else [])
        out = json.loads(subprocess.check_output(args))
    finally:
        os.unlink(f.name)
    assert out["selected"] and out["restored"]
    oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    want, _, _ = cabi.decode_stream(oracle, es)                     # checker: md5(Y|Cr|Cb) per picture
    assert out["video"] == want
    assert out["sizes"] == [[176, 144]] and abs(out["frameRate"] - 30.0) < 1e-6
    assert out["audio"] == afx["frame_md5"] and out["sampleRate"] == afx["sample_rate"]


@pytest.mark.gpu
def test_dropped_handles_give_their_decoders_back(hip_lib):
    """A handle that is garbage-collected without destroy() must neither free memory its plane views still look at nor
    keep the decoder for ever: the decoder goes with the last of {handle, external ArrayBuffers} (napi_addon.c, dec_owner_t)."""
    build.build_addon()
    fx, ts = _ts_for("cfg0_240p_intra")
    try:
        out = json.loads(subprocess.check_output([NODE, "--expose-gc", os.path.join(ROOT, "tests", "js", "hip_handle_lifetime.js"), ts]))
    finally:
        os.unlink(ts)
    assert out["start"] == 0 and out["afterCreate"] == 12
    assert out["afterDrop"] == 0, out                       # nothing leaks
    assert out["viewsStillReadable"], out                   # nothing dangles
    assert out["viewsKeepDecoder"] in (0, 1), out           # 1: zero-copy views (external buffers); 0: the host copies planes
    assert out["afterViewsGone"] == 0 and out["afterDestroy"] == 0, out
    assert out["detachedLength"] == 0 or out["viewsKeepDecoder"] == 0, out


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["static", "streaming"])
def test_the_references_own_player_drives_the_hip_classes_on_the_gpu(mode, hip_lib):
    """The reference's own JSMpeg.Player, Demuxer.TS, Decoder.Base and Canvas2D renderer -- from its shipped bundle
    (oracle/_ref/jsmpeg_ref.min.js, placed there unmodified by oracle/Makefile; it travels to the GPU box) -- drive
    MPEG1VideoHIP / MP2AudioHIP over the real addon (JSMpeg.PlayerHIP), and in the SAME Node process the bundle's own
    decoders check them live: the event log (every Canvas2D frame, every audio buffer and its start time, the clock,
    the seek) equals the reference Player's with its wasm decoders, and every rendered (Y, Cr, Cb) equals both the wasm
    decoder's and the pure-JS JSMpeg.Decoder.MPEG1Video's (reference src/player.js:30-46, 195-294)."""
    bundle = build.JS_REF
    if not os.path.exists(bundle):
        pytest.skip("oracle/_ref/jsmpeg_ref.min.js not there (made from /root/reference by oracle/Makefile)")
    from test_mp2_gpu import _av_ts
    build.build_addon()
    ts, es, afx, _ = _av_ts(30, "stereo_44k_192", 3)
    f = tempfile.NamedTemporaryFile(suffix=".ts", delete=False)
    f.write(ts.tobytes())
    f.close()
    try:
        args = [NODE, os.path.join(ROOT, "tests", "js", "player_bundle_gpu.js"), bundle, f.name] + (["streaming"] if mode == "streaming" else [])
        out = json.loads(subprocess.check_output(args, timeout=300))
    finally:
        os.unlink(f.name)
    assert out["selected"] and out["restored"] and out["realParts"] and out["referenceRunsUsedWasmAndJs"], out
    assert out["sameLogAsWasmPlayer"], out
    assert out["samePlanesAsWasm"] and out["samePlanesAsJsDecoder"], out
    assert out["frames"] >= 15 and out["audio"] >= 10 and out["pictures"] >= 15


def test_router_probes_the_picture_size_on_the_host():
    """HIPBatchRouter.probeTS / probeES read the 12 + 12 bits behind the first 00 00 01 B3 (mpeg1.c:872-880) out of the first
    packets / bytes -- routing, not decoding; no GPU needed"""
    script = os.path.join(ROOT, "tests", "js", "_probe_tmp.js")
    out = []
    for (w, h) in ((176, 144), (1920, 1080), (17, 33)):
        es, offs = synth.generate_config("cfg1_720p", n_frames=2, width=w, height=h)
        with tempfile.NamedTemporaryFile(suffix=".ts", delete=False) as f, tempfile.NamedTemporaryFile(suffix=".es", delete=False) as g:
            f.write(synth.mux_ts(es, offs).tobytes())
            g.write(es.tobytes())
        code = ("const {install}=require(%r);const {HIPBatchRouter}=install({}, {binding:{}});const fs=require('fs');"
                "console.log(JSON.stringify([HIPBatchRouter.probeTS(fs.readFileSync(%r)),HIPBatchRouter.probeES(fs.readFileSync(%r)),HIPBatchRouter.probeES(new Uint8Array(64))]))"
                % (os.path.join(ROOT, "jsmpeg_amd", "js", "batch-hip.js"), f.name, g.name))
        try:
            out.append(json.loads(subprocess.check_output([NODE, "-e", code])))
        finally:
            os.unlink(f.name)
            os.unlink(g.name)
    assert out == [[{"width": w, "height": h}, {"width": w, "height": h}, None] for (w, h) in ((176, 144), (1920, 1080), (17, 33))]


@pytest.mark.gpu
def test_node_router_mixed_geometries_and_the_decoders_clock(hip_lib):
    """five TS files of three picture sizes in one decodeTS call: JSMpeg.HIPBatchRouter keeps a HIPBatch per size, every frame
    comes back under ITS buffer's index with its own size and pts, bit-exact to the oracle; and elementary streams (no time
    stamps) get the decoder's own clock: 1 / frameRate of the stream's sequence header per picture (mpeg1.js:57) -- here 25 fps"""
    build.build_addon()
    sizes = [(352, 288), (176, 144), (352, 288), (320, 192), (176, 144)]
    paths, want = [], []
    es25 = None
    try:
        for s, (w, h) in enumerate(sizes):
            es, offs = synth.generate_config("cfg1_720p", n_frames=7, stream=70 + s, width=w, height=h)
            f = tempfile.NamedTemporaryFile(suffix=".ts", delete=False)
            f.write(synth.mux_ts(es, offs).tobytes())
            f.close()
            paths.append(f.name)
            want.append(cabi_md5_frames(es))
            if s == 0:
                es25 = es.copy()
                assert bytes(es25[:4]) == b"\x00\x00\x01\xb3"
                es25[7] = (es25[7] & 0xF0) | 3                   # picture_rate code 3 = 25 fps (mpeg1.c:988-991); only the first header counts
        g = tempfile.NamedTemporaryFile(suffix=".m1v", delete=False)
        g.write(es25.tobytes())
        g.close()
        paths.append(g.name)
        out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "hip_router.js"), "--es", g.name] + paths[:-1], timeout=300))
    finally:
        for p in paths:
            os.unlink(p)
    assert out["frames"] == 5 * 7 and out["batches"] == ["176x144", "320x192", "352x288"] and out["skipped"] == []
    for s, (w, h) in enumerate(sizes):
        st = out["streams"][s]
        assert st["planes"] == want[s] and st["sizes"] == [[w, h]] * 7
        assert abs(st["pts"][0] - 0.1) < 1e-6 and abs(st["pts"][3] - st["pts"][2] - 1 / 30) < 1e-4
    assert out["esFrames"] == 7 and out["esPlanes"] == want[0]
    assert out["esPts"] == [round(k / 25.0, 6) for k in range(7)]
    assert out["skippedLater"] == [0]


def test_decode_async_class_logic_over_an_injected_binding():
    """JSMpeg.HIPBatch.decodeAsync: the picture count arrives with the promise; while it is pending every other call on the batch
    throws (a batch object is one thread's at a time) and a second decodeAsync is rejected; afterwards the batch is usable again;
    a failed decode rejects and frees the batch too.  (The binding is injected: no addon, no GPU.)"""
    script = r"""
const { install } = require(%r);
let resolveDecode, rejectDecode, destroyed = 0;
const binding = {
  batchCreate() { return {}; }, batchGeometry() { return { codedWidth: 16, codedHeight: 16, lumaBytes: 256, chromaBytes: 64 }; },
  batchUpload() {}, batchDecode() { return 3; }, batchDestroy() { destroyed++; },
  batchDecodeAsync() { return new Promise((res, rej) => { resolveDecode = res; rejectDecode = rej; }); },
  batchFrameHashes() {}, hostUnregister() {},
};
const { HIPBatch } = install({}, { binding });
const out = {};
const threw = (f) => { try { f(); return false; } catch (e) { return /in flight/.test(e.message); } };
(async () => {
  const b = new HIPBatch({ width: 16, height: 16 });
  b.upload([new Uint8Array(4)]);
  const p = b.decodeAsync();
  out.busy = b.decoding === true;
  out.guards = [threw(() => b.decode()), threw(() => b.upload([])), threw(() => b.destroy()), threw(() => b.frameHashes()), threw(() => b.readPlanes(0))];
  out.second = await b.decodeAsync().then(() => 'resolved', (e) => /in flight/.test(e.message) ? 'rejected' : 'other');
  resolveDecode(7);
  out.n = await p;
  out.after = [b.decoding, b.pictures, b.decode()];
  const q = b.decodeAsync();
  rejectDecode(new Error('device lost'));
  out.failed = await q.then(() => 'resolved', (e) => e.message);
  out.freed = b.decoding === false;
  b.destroy();
  out.destroyed = destroyed;
  console.log(JSON.stringify(out));
})();
""" % os.path.join(ROOT, "jsmpeg_amd", "js", "batch-hip.js")
    out = json.loads(subprocess.check_output([NODE, "-e", script]))
    assert out == {"busy": True, "guards": [True] * 5, "second": "rejected", "n": 7, "after": [False, 7, 3], "failed": "device lost",
                   "freed": True, "destroyed": 1}


@pytest.mark.gpu
def test_two_batches_in_flight_from_node(hip_lib):
    """tools/bench_node.js --two: two HIPBatch objects, a chain of decodeAsync() each (napi_async_work), side by side on one GPU; the
    pictures of both frame pools against the oracle's hashes"""
    from jsmpeg_amd import cabi, hashing
    build.build_addon()
    lib = build.build_oracle()
    streams = [synth.generate_config("cfg1_720p", n_frames=12, width=352, height=288, stream=s, gop=6)[0] for s in range(6)]
    want = {}
    for s, es in enumerate(streams):
        frames = cabi.decode_stream(lib, es, keep="planes")[0]
        want[str(s)] = ["%016x" % hashing.frame_hash(*f) for f in frames]
    with tempfile.TemporaryDirectory() as td:
        for i, es in enumerate(streams):
            es.tofile(os.path.join(td, "s%d.m1v" % i))
        hp = os.path.join(td, "hashes.json")
        json.dump(want, open(hp, "w"))
        out = subprocess.check_output([NODE, os.path.join(ROOT, "tools", "bench_node.js"), "--dir", td, "--streams", "6", "--width", "352", "--height", "288",
                                       "--frames", "12", "--steps", "3", "--warmup", "1", "--hashes", hp, "--two", "80"], timeout=300)   # (80 passes: the chains are long against the timer that staggers them)
    res = json.loads([ln for ln in out.decode().splitlines() if ln.startswith("{")][-1])
    assert "error" not in res, res
    two = res["two_batches_in_flight"]
    assert "error" not in two, two
    assert two["parity"].startswith("every picture of both frame pools") and two["passes_in_window"] >= 80 and two["value"] > 0
