"""The Node.js host side of the MP2 stage: N-API addon (mp2* bindings) + JSMpeg.Decoder.MP2AudioHIP."""
import json
import os
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT, have_reference
from jsmpeg_amd import build
from mp2_util import FIXTURES, FIXTURE_IDS, load_case
from ts_craft import Muxer

NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node not installed")


def _audio_ts(case, frames_per_pes=2):
    """The fixture's frames as an MPEG-TS file: audio PES packets (stream 0xC0, PES_packet_length set, pts from
    the frame number), a video-looking PID and null packets in between."""
    fx, data, offs = load_case(FIXTURES[FIXTURE_IDS.index(case)])
    m = Muxer()
    for k in range(0, fx["n_frames"], frames_per_pes):
        hi = min(k + frames_per_pes, fx["n_frames"])
        pts = 90000 + int(round(90000 * 1152 * k / fx["sample_rate"]))
        m.pes(0x101, 0xC0, data[int(offs[k]):int(offs[hi])].tobytes(), pts=pts, with_length=True)
        if k % 4 == 0:
            m.packet(0x1fff, b"")
    f = tempfile.NamedTemporaryFile(suffix=".ts", delete=False)
    f.write(m.bytes().tobytes())
    f.close()
    return fx, f.name


def test_addon_exports_the_mp2_abi():
    addon = build.build_addon()
    out = subprocess.check_output([NODE, "-e", "const a=require(%r);console.log(JSON.stringify(Object.keys(a)))" % addon])
    assert {"mp2Create", "mp2Destroy", "mp2BufferWrite", "mp2GetIndex", "mp2SetIndex", "mp2GetSampleRate", "mp2Decode",
            "mp2GetChannels", "mp2BatchCreate", "mp2BatchDestroy", "mp2BatchUpload", "mp2BatchUploadTS", "mp2BatchDecode",
            "mp2BatchFrameCount", "mp2BatchFrameInfo", "mp2BatchTsWrites", "mp2BatchReadPCM"} <= set(json.loads(out))


def test_class_fails_loudly_without_gpu():
    from conftest import have_gpu
    if have_gpu():
        pytest.skip("a GPU is present")
    build.build_addon()
    script = ("const {install}=require(%r);const {MP2AudioHIP}=install();const d=new MP2AudioHIP({});"
              "console.log(d.decode());"
              "try{d.write(0,[new Uint8Array(8)]);console.log('NO THROW')}catch(e){console.log('THROWS:'+e.message)}"
              % os.path.join(ROOT, "jsmpeg_amd", "js", "mp2-hip.js"))
    out = subprocess.check_output([NODE, "-e", script]).decode().splitlines()
    assert out[0] == "false" and out[1].startswith("THROWS:") and "no CPU fallback" in out[1]


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference")
@pytest.mark.parametrize("mode", ["static", "streaming"])
@pytest.mark.parametrize("case", ["varying_44k", "mono_32k_48"])
def test_class_is_a_dropin_for_the_wasm_wrapper(case, mode):
    """Same play / onAudioDecode / currentTime / index / seek event sequence as the reference's
    JSMpeg.Decoder.MP2AudioWASM, driven by the reference's own TS demuxer (binding = the reference's wasm exports,
    so only the class is under test)."""
    fx, ts = _audio_ts(case)
    try:
        args = [NODE, os.path.join(ROOT, "tests", "js", "mp2_class_vs_reference.js"), ts]
        if mode == "streaming":
            args.append("streaming")
        out = json.loads(subprocess.check_output(args))
    finally:
        os.unlink(ts)
    assert out["same"], out
    assert out["plays"] >= fx["n_frames"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["static", "streaming"])
@pytest.mark.parametrize("case", ["varying_44k", "joint_48k_128"])
def test_node_class_on_gpu_matches_golden(case, mode, hip_lib):
    """TS file -> ts-demux -> MP2AudioHIP -> addon -> HIP kernels: every frame's samples bit for bit."""
    build.build_addon()
    fx, ts = _audio_ts(case, frames_per_pes=3)
    try:
        args = [NODE, os.path.join(ROOT, "tests", "js", "hip_mp2_decode.js"), ts]
        if mode == "streaming":
            args.append("streaming")
        out = json.loads(subprocess.check_output(args))
    finally:
        os.unlink(ts)
    assert out["frames"] == fx["frame_md5"]
    assert out["sampleRate"] == (fx["sample_rate"] if case != "varying_44k" else 44100)
    assert len(out["indices"]) == fx["n_frames"]
    if mode == "static":
        assert out["indices"] == fx["bit_index_after_decode"]


@pytest.mark.gpu
def test_node_batch_audio_and_video_from_the_same_ts(hip_lib, libs):
    """TS files with a video and an MP2 audio stream -> JSMpeg.HIPBatch({audio: true}): pictures against the oracle,
    PCM against the audio fixtures, audio time stamps as the reference's Decoder.Base assigns them (pts of the PES a
    frame starts in for the first such frame, + 1152 / rate after that)."""
    import hashlib
    from jsmpeg_amd import cabi
    from test_mp2_gpu import _av_ts
    build.build_addon()
    cases = [_av_ts(9, "stereo_44k_192", 3), _av_ts(6, "mono_32k_48", 4)]
    paths = []
    try:
        for c in cases:
            f = tempfile.NamedTemporaryFile(suffix=".ts", delete=False)
            f.write(c[0].tobytes())
            f.close()
            paths.append(f.name)
        out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "hip_batch_av.js"), "176", "144"] + paths))
    finally:
        for p in paths:
            os.unlink(p)
    assert out["pictures"] == 15 and out["audioFrames"] == sum(c[2]["n_frames"] for c in cases)
    for s, (ts, es, fx, data) in enumerate(cases):
        assert out["streams"][s]["planes"] == cabi.decode_stream(libs["oracle"], es)[0]
        assert out["streams"][s]["audio"] == fx["frame_md5"]
        assert out["streams"][s]["sampleRate"] == fx["sample_rate"]
        pts = out["streams"][s]["audioPts"]
        for k in range(fx["n_frames"]):            # two frames per PES, pts of the PES = 1 + 1152 k / rate
            base = 1.0 + int(90000 * 1152 * (k - k % 2) / fx["sample_rate"]) / 90000.0
            assert abs(pts[k] - (base + (k % 2) * 1152 / fx["sample_rate"])) < 1e-9, (s, k)


def test_batch_audio_time_stamps_follow_decoder_base():
    """HIPBatch.forEachAudioFrame over a stand-in binding (no GPU): a frame that starts a PES gets that PES's pts, the
    frames after it 1152 / rate each (what Decoder.Base.advanceDecodedTime does, src/decoder.js:73-93); a PES that
    starts in the middle of a frame does not restamp that frame."""
    script = r"""
const { install } = require(%r);
const frames = [ {byteOffset: 0}, {byteOffset: 600}, {byteOffset: 1200}, {byteOffset: 1800}, {byteOffset: 2400} ];
const writes = [ {pts: 1.0, offset: 0, length: 1200}, {pts: 2.0, offset: 1200, length: 900}, {pts: 3.0, offset: 2100, length: 900} ];
const binding = {
  batchCreate: () => ({}), batchGeometry: () => ({ codedWidth: 16, codedHeight: 16, lumaBytes: 256, chromaBytes: 64 }), batchDestroy() {},
  mp2BatchCreate: () => ({}), mp2BatchDestroy() {}, mp2BatchUploadTS() {}, mp2BatchTsWrites: () => writes, mp2BatchDecode: () => frames.length,
  mp2BatchFrameCount: () => frames.length, mp2BatchFrameInfo: (h, s, f) => ({ byteOffset: frames[f].byteOffset, byteSize: 600, sampleRate: 48000 }),
  mp2BatchReadPCM: (h, s, first, count, pcm) => { for (let i = 0; i < count * 2304; i++) pcm[i] = first + i / 2304; return count; },
};
const { HIPBatch } = install(null, { binding });
const b = new HIPBatch({ width: 16, height: 16, audio: true });
b.uploadAudioTS([new Uint8Array(188)]);
b.decodeAudio();
const out = [];
b.forEachAudioFrame((a) => out.push([a.stream, a.index, +a.pts.toFixed(6), a.sampleRate, a.left.length, a.right.length, Math.floor(a.left[0]), Math.floor(a.right[0])]));
console.log(JSON.stringify(out));
""" % os.path.join(ROOT, "jsmpeg_amd", "js", "batch-hip.js")
    out = json.loads(subprocess.check_output([NODE, "-e", script]))
    d = 1152 / 48000
    # frames 0 and 2 start a PES; the third PES begins inside frame 3 (1800 < 2100 < 2400), so frame 3 runs on from
    # frame 2 and frame 4 -- the first to start inside that PES -- takes its pts
    want_pts = [1.0, 1.0 + d, 2.0, 2.0 + d, 3.0]
    assert [o[:2] for o in out] == [[0, i] for i in range(5)]
    assert all(o[3] == 48000 and o[4] == 1152 and o[5] == 1152 for o in out)
    assert [o[2] for o in out] == [round(x, 6) for x in want_pts]
    assert [o[6] for o in out] == [0, 1, 2, 3, 4]
