"""The Node.js host of the LIVE AUDIO streams: jsmpeg_amd/js/live-audio-hip.js (JSMpeg.HIPLiveAudio) over napi_live_audio.c.
CPU: the class logic over an injected binding, the addon's exports, loud failure without a GPU.  GPU: TS files with a video
and an audio stream through ONE demuxer per file (the reference's own Demuxer.TS from its shipped bundle where that is there)
into a live video stream and a live audio stream, a tick of each per round of writes; every picture and every frame of samples
against the oracle / the golden fixtures."""
import json
import os
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT
from jsmpeg_amd import build, cabi

NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node not installed")


def test_addon_exports_the_live_audio_functions():
    addon = build.build_addon()
    out = subprocess.check_output([NODE, "-e", "const a=require(%r);console.log(JSON.stringify(Object.keys(a)))" % addon])
    assert {"liveAudioCreate", "liveAudioDestroy", "liveAudioOpen", "liveAudioClose", "liveAudioWrite", "liveAudioWriteTS", "liveAudioTick",
            "liveAudioFrame", "liveAudioReadPCM", "liveAudioStreamInfo", "liveAudioTimings"} <= set(json.loads(out))


def test_live_audio_class_fails_loudly_without_gpu():
    from conftest import have_gpu
    if have_gpu():
        pytest.skip("a GPU is present")
    build.build_addon()
    script = ("const {install}=require(%r);const {HIPLiveAudio}=install();"
              "try{new HIPLiveAudio({});console.log('NO THROW')}catch(e){console.log('THROWS:'+e.message)}"
              % os.path.join(ROOT, "jsmpeg_amd", "js", "live-audio-hip.js"))
    out = subprocess.check_output([NODE, "-e", script]).decode()
    assert out.startswith("THROWS:") and "no CPU fallback" in out


def test_live_audio_class_logic_over_an_injected_binding():
    """a stream has the decoder's surface (reference src/decoder.js:3-106, src/mp2-wasm.js:55-115): write copies through as ONE
    write; play(sampleRate, left, right) per frame with views of 1152 samples into ONE read of the tick's samples; onAudioDecode;
    decodedTime += 1152 / sampleRate; currentTime less what the output holds; the Player's catching-up rule (player.js:232-241):
    an output more than maxAudioLag behind is reset and muted for the stream's frames of this tick, then switched on again"""
    out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "live_audio_class_fake.js")]))
    assert out["calls"] == [["liveAudioCreate", 2, 3, 4096, 1], ["liveAudioOpen"], ["liveAudioOpen"], ["liveAudioWrite", 0, 0.5, 626], ["liveAudioTick"],
                            ["liveAudioTick"], ["liveAudioReadPCM", 0, 1, True], ["liveAudioWriteTS", 1, 188, 192], ["liveAudioTick"],
                            ["liveAudioReadPCM", 0, 3, True], ["liveAudioTick"], ["liveAudioClose", 1], ["liveAudioDestroy"]]
    assert out["log"] == [["tick", 0], ["play", 44100, 10, -10, 1152, 1152, True], ["decoded", 0], ["frame", 0, 0, 0.5, 44100, 1152, 626], ["tick", 1],
                          ["play", 44100, 20, -20, 1152, 1152, False], ["decoded", 0], ["frame", 0, 1, 0.5, 44100, 20, -20],
                          ["play", 44100, 21, -21, 1152, 1152, False], ["decoded", 0], ["frame", 0, 2, 0.6, 44100, 21, -21], ["frame", 1, 0, 7, 32000, 22, -22],
                          ["tick", 3], ["state", 44100, 0.078367, 0.078367, True, 626, 3, 32000, 0.036, 1880, True, True, 1], ["decode", False, 0],
                          ["closedThrows", True, 1]]


@pytest.mark.gpu
@pytest.mark.parametrize("demuxer", ["ts-demux.js", "reference bundle"])
def test_node_live_audio_and_video_from_one_demuxer_on_gpu(demuxer, hip_lib):
    """3 TS files (video 0xE0 + audio 0xC0, two audio frames per PES) -> one demuxer per file feeding a HIPLive stream and a
    HIPLiveAudio stream, ragged pieces, a tick of each per round: every rendered picture == the oracle's, every played frame of
    samples == the golden fixture's, sample rates and decoded time as the reference's decoder would report them"""
    from test_mp2_gpu import _av_ts
    build.build_addon()
    oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    extra = []
    if demuxer == "reference bundle":
        if not os.path.exists(build.JS_REF):
            pytest.skip("oracle/_ref/jsmpeg_ref.min.js not there (made from /root/reference by oracle/Makefile)")
        extra = ["--bundle", build.JS_REF]
    cases = [_av_ts(9, "stereo_44k_192", 3), _av_ts(6, "mono_32k_48", 4), _av_ts(12, "varying_44k", 5)]
    paths = []
    for c in cases:
        f = tempfile.NamedTemporaryFile(suffix=".ts", delete=False)
        f.write(bytes(c[0]))
        f.close()
        paths.append(f.name)
    try:
        out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "hip_live_av.js"), "176", "144"] + extra + ["--packets", "9"] + paths,
                                                 timeout=300))
    finally:
        for p in paths:
            os.unlink(p)
    assert out["pictures"] == 9 + 6 + 12 and out["frames"] == sum(c[2]["n_frames"] for c in cases)
    for s, (tsb, es, fx, data) in enumerate(cases):
        st = out["streams"][s]
        assert st["pcm"] == fx["frame_md5"], s
        assert st["planes"] == cabi.decode_stream(oracle, es)[0], s
        assert st["rates"] == [fx["sample_rate"]] * fx["n_frames"] and st["audioCallbacks"] == fx["n_frames"]
        assert out["sampleRates"][s] == fx["sample_rate"] and abs(out["decodedTimes"][s] - fx["n_frames"] * 1152 / fx["sample_rate"]) < 1e-6
        assert out["pending"][s] == 0 and out["audioBytes"][s] == len(data)
        # two frames per PES: a frame's pts is its PES's
        want_pts = [1.0 + 1152 * (k - k % 2) / fx["sample_rate"] for k in range(fx["n_frames"])]
        assert all(abs(a - b) < 2e-5 for a, b in zip(st["audioPts"], want_pts))
    assert out["rounds"] > 5
