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
            "lastError"} <= names


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
