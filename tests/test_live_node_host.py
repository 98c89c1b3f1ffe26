"""The Node.js host of the LIVE-stream interface: jsmpeg_amd/js/live-hip.js (JSMpeg.HIPLive) over napi_live.c.
CPU: the class logic over an injected binding, the addon's exports, loud failure without a GPU.  GPU: TS files through a
demuxer per stream (the reference's own Demuxer.TS from its shipped bundle where that is there) into live streams, a tick
per round of writes, every picture against the oracle."""
import hashlib
import json
import os
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT
from jsmpeg_amd import build, cabi, synth

NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node not installed")


def test_addon_exports_the_live_functions():
    addon = build.build_addon()
    out = subprocess.check_output([NODE, "-e", "const a=require(%r);console.log(JSON.stringify(Object.keys(a)))" % addon])
    assert {"liveCreate", "liveDestroy", "liveOpen", "liveClose", "liveWrite", "liveWriteTS", "liveTick", "liveTickBegin", "liveTickEnd", "livePicture", "liveReadFrames", "hostRegister", "hostUnregister", "liveReadPlanes", "liveReadRGBA",
            "liveFrameHashes", "liveStreamInfo", "liveGeometry", "liveTimings"} <= set(json.loads(out))


def test_live_class_fails_loudly_without_gpu():
    from conftest import have_gpu
    if have_gpu():
        pytest.skip("a GPU is present")
    build.build_addon()
    script = ("const {install}=require(%r);const {HIPLive}=install();"
              "try{new HIPLive({width:320,height:240});console.log('NO THROW')}catch(e){console.log('THROWS:'+e.message)}"
              % os.path.join(ROOT, "jsmpeg_amd", "js", "live-hip.js"))
    out = subprocess.check_output([NODE, "-e", script]).decode()
    assert out.startswith("THROWS:") and "no CPU fallback" in out


def test_live_class_logic_over_an_injected_binding():
    """a stream has the decoder's surface (reference src/decoder.js:3-106, src/mpeg1-wasm.js:72-128): write copies through as
    ONE write, the header is polled after the tick that saw it -> destination.resize once, render(y, cr, cb, false) per
    picture, onVideoDecode, decodedTime += 1 / frameRate; tick() passes flush on, hands out frames, RGBA on request.
    The tick in two halves: a second tickBegin throws, writes between the halves go through and the library is asked nothing
    else meanwhile (any other call would end the tick), writeTS's byte count is caught up at tickEnd, tickAsync resolves a
    turn of the event loop later"""
    out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "live_class_fake.js")]))
    assert out["calls"] == [["liveCreate", 30, 15, 2, 3, 4096, 1], ["liveOpen"], ["liveOpen"], ["liveWrite", 0, 0.5, 140], ["liveTick", True],
                            ["liveTick", True], ["liveWrite", 1, 7, 8], ["liveTick", False], ["liveReadRGBA", 1, 30 * 15 * 4],
                            ["liveTickBegin", True], ["liveWrite", 1, 9, 5], ["liveWriteTS", 1, 188, 224], ["liveTickEnd"], ["liveTickBegin", True], ["liveClose", 1],
                            ["liveTickEnd"], ["liveDestroy"]]
    assert out["log"] == [["tick", 0], ["resize", 30, 15], ["render", 10, 1, 2, False, 512, 128], ["decoded", 0], ["frame", 0, 0, 0.5, 1, True], ["tick", 1],
                          ["render", 20, 1, 2, False, 512, 128], ["decoded", 0], ["frame", 0, 1, 0.6, 2, None, 20], ["frame", 1, 0, 7, 1, 99, None], ["tick", 2],
                          ["hash", "01000000000000ef"], ["state", True, 25, 30, 15, 512, 0.08, True, 140, True, 0.04], ["decode", False],
                          ["beside", True, True, 13], ["frame", 1, 1, 8], ["tickEnd", 1, False, 100, 0], ["closedThrows", True, 1]]
    assert out["later"] == [["begun", True], ["async", 0, False]]
    # the planes of a tick's pictures come in ONE call into one array, pinned once, unpinned at destroy; frames are views into it
    assert out["reads"] == [[0, 1, 768, 768], [0, 1, 768, 768]] and out["pins"] == [["pin", 768], ["unpin", 768]]


def test_live_router_logic_over_an_injected_binding():
    """JSMpeg.HIPLiveRouter: a stream holds its writes until its first sequence header shows (also one cut by a write), joins
    the HIPLive of that size -- made on demand, one per size -- and replays what it held in order; ticks reach every size,
    frames name the router's stream"""
    out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "live_router_fake.js")]))
    assert out["calls"] == [["liveCreate", 32, 16, 7], ["liveOpen", 0], ["liveWrite", 0, 0, 1, 3, 9], ["liveWrite", 0, 0, 2, 5, 0], ["liveWrite", 0, 0, 3, 8, 0],
                            ["liveCreate", 48, 32, 7], ["liveOpen", 1], ["liveWrite", 1, 0, 4, 12, 0], ["liveOpen", 0], ["liveWrite", 0, 1, 5, 12, 0],
                            ["liveWrite", 0, 0, 6, 1, 5], ["liveTick", 0], ["liveTick", 1], ["liveClose", 0, 1], ["liveDestroy", 0], ["liveDestroy", 1]]
    assert out["log"] == [["held", False, 3, 3, 0], ["resize a", 32, 16], ["bound", True, 32, 16, 48, 32, 0, 0, 1, 2, 0],
                          ["frame", "c", 32, 16, 5], ["render a"], ["frame", "a", 32, 16, 3], ["frame", "b", 48, 32, 4], ["tick", 3]]


def test_live_pipelined_class_logic_over_an_injected_binding():
    """HIPLive({pipelined: true}): a tick ends the read-out of the tick before, starts its own pictures on their way into the OTHER
    pinned array, then hands out the tick before's frames; a tick without pictures still hands out the one before; a stream closed
    meanwhile gets nothing; drain() the last ones; both arrays unpinned at destroy; the plain liveReadFrames is never called"""
    out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "live_pipelined_fake.js")]))
    assert out["calls"] == [["liveTick"], ["pin", 0], ["begin", 0, 2, 0, 768], ["liveTick"], ["end"], ["pin", 1], ["begin", 0, 1, 1, 768], ["liveClose", 1],
                            ["liveTick"], ["end"], ["liveTick"], ["begin", 0, 1, 0, 768], ["end"], ["unpin"], ["unpin"], ["liveDestroy"]]
    assert out["log"] == [["tick", 0], ["frame", 0, 0, 1, 0, 512, 128], ["frame", 1, 0, 1.5, 1, 512, 128], ["tick", 2], ["frame", 0, 1, 2, 10, 512, 128], ["tick", 1],
                          ["tick", 0], ["drain", 0, 0], ["state", 2, 0.08]]


def test_live_router_spreads_streams_over_devices():
    """JSMpeg.HIPLiveRouter({devices: [...]}): streams shard by stream over the GPUs of a node with no exchange -- a handle per
    (size, device) made on demand, a new stream joins the device that holds the fewest, a full device is passed over, and a tick puts
    every handle's pass on its device (tickBegin) before it waits for the first (tickEnd)"""
    out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "live_router_devices_fake.js")]))
    assert out["where"] == [4, 5, 6, 4, 5, 6, 4] and out["late"] == 6 and out["full"] is True
    assert out["creates"] == [["liveCreate", 32, 16, 4], ["liveCreate", 32, 16, 5], ["liveCreate", 32, 16, 6], ["liveCreate", 48, 32, 6], ["liveCreate", 48, 32, 4]]
    assert out["tickCalls"] == ["liveTickBegin"] * 5 + ["liveTickEnd"] * 5
    assert out["frames"] == [[4, 32], [5, 32], [6, 32], [4, 48]]


def _ts_files(n, frames, w, h):
    paths, want, es_all = [], [], []
    oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    for s in range(n):
        kw = dict(mv_jitter=1, f_code_max=1, coded_permille=60, ac_max=1, gop=2) if s == 1 else {}     # stream 1: unwritten last macroblocks
        es, offs = synth.generate_config("cfg1_720p", n_frames=frames, stream=40 + s, width=w, height=h, **kw)
        f = tempfile.NamedTemporaryFile(suffix=".ts", delete=False)
        f.write(synth.mux_ts(es, offs).tobytes())
        f.close()
        paths.append(f.name)
        want.append(cabi.decode_stream(oracle, es)[0])            # checker: md5(Y|Cr|Cb) per picture
        es_all.append(es)
    return paths, want, es_all


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [False, True], ids=["tick", "pieces_written_beside_tickAsync"])
@pytest.mark.parametrize("demuxer", ["ts-demux.js", "reference bundle", "the library's own (writeTS)"])
def test_node_live_streams_on_gpu(demuxer, overlap, hip_lib):
    """4 TS files -> a demuxer per stream -> JSMpeg.HIPLive over the real addon, fed in ragged pieces round-robin, a tick per
    round; the last stream joins 5 rounds late.  Every rendered picture of every stream == the oracle's; pts as the demuxer
    reported them; one resize per stream.  Second form: a round's pieces are written while the tick of the round before is
    on the device (live.tickAsync = tickBegin, a turn of the event loop, tickEnd): the same pictures in the same rounds."""
    build.build_addon()
    extra = []
    if demuxer == "reference bundle":
        if not os.path.exists(build.JS_REF):
            pytest.skip("oracle/_ref/jsmpeg_ref.min.js not there (made from /root/reference by oracle/Makefile)")
        extra = ["--bundle", build.JS_REF]
    if demuxer.startswith("the library"):
        extra = ["--native-ts"]
    if overlap:
        extra.append("--overlap")
    paths, want, _ = _ts_files(4, 14, 352, 288)
    try:
        out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "hip_live_ts.js"), "352", "288"] + extra +
                                                 ["--late", "5", "--packets", "24"] + paths, timeout=300))
    finally:
        for p in paths:
            os.unlink(p)
    assert out["pictures"] == 4 * 14 == out["hashesSeen"]
    for s in range(4):
        st = out["streams"][s]
        assert st["planes"] == want[s], s
        assert st["sizes"] == [[352, 288]] and st["callbacks"] == 14
        assert len(st["pts"]) == 14 and abs(st["pts"][0] - 0.1) < 1e-6 and abs(st["pts"][5] - st["pts"][4] - 1 / 30) < 1e-4
        assert st["types"] == [1 if k % (2 if s == 1 else 12) == 0 else 2 for k in range(14)]
        assert abs(out["decodedTimes"][s] - 14 / 30) < 1e-5 and abs(out["frameRates"][s] - 30) < 1e-6
    assert out["pending"] == [0] * 4 and out["evictions"] == [0] * 4 and out["closedStreamThrows"]
    assert out["overlap"] == overlap and out["bytesWritten"] == out["bytesWrittenInfo"]
    assert out["rounds"] > 14                                   # (pieces of ~24 packets: pictures arrive over several rounds)


@pytest.mark.gpu
def test_node_live_pipelined_read_out(hip_lib):
    """HIPLive({pipelined: true}): a tick hands out the pictures of the tick BEFORE it -- their planes went to the host beside this
    tick's pass (jsmpeg_hip_live_read_frames_begin / _end, two pinned arrays in turn) -- and drain() the last ones: the same
    pictures in the same order as the plain form, one tick later"""
    build.build_addon()
    paths, want, _ = _ts_files(4, 14, 352, 288)
    try:
        out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "hip_live_ts.js"), "352", "288", "--pipelined", "--late", "3",
                                                  "--packets", "24"] + paths, timeout=300))
    finally:
        for p in paths:
            os.unlink(p)
    assert out["pipelined"] is True and out["pictures"] == 4 * 14 == out["hashesSeen"]
    for s in range(4):
        st = out["streams"][s]
        assert st["planes"] == want[s], s
        assert st["sizes"] == [[352, 288]] and st["callbacks"] == 14 and len(st["pts"]) == 14
        assert abs(out["decodedTimes"][s] - 14 / 30) < 1e-5


@pytest.mark.gpu
def test_node_live_rgba_frames(hip_lib):
    """tick({rgba: true}): frames as Canvas2D-identical RGBA, converted on the device"""
    import numpy as np
    from oracle import checkers
    build.build_addon()
    paths, _, es_all = _ts_files(2, 5, 352, 288)
    oracle = build.LIB_ORACLE
    want = []
    for es in es_all:
        frames, _, _ = cabi.decode_stream(oracle, es, keep="planes")
        want.append([hashlib.md5(np.ascontiguousarray(checkers.oracle_rgba(oracle, *f, 352, 288)).tobytes()).hexdigest() for f in frames])
    try:
        out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "hip_live_ts.js"), "352", "288", "--rgba"] + paths, timeout=300))
    finally:
        for p in paths:
            os.unlink(p)
    assert [st["rgba"] for st in out["streams"]] == want


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, "0,0"], ids=["one handle per size", "two handles per size on GPU 0 (the N-GPU form rehearsed)"])
def test_node_live_router_streams_of_several_sizes(hip_lib, devices):
    """JSMpeg.HIPLiveRouter over the real addon: five TS files of three picture sizes, fed in ragged pieces (ts-demux.js ->
    write() for the even ones, writeTS for the odd ones); the router finds each stream's size in the stream, one HIPLive per
    size; every rendered picture == the oracle's"""
    build.build_addon()
    oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    sizes = [(352, 288), (176, 144), (352, 288), (320, 192), (176, 144)]
    paths, want = [], []
    for s, (w, h) in enumerate(sizes):
        es, offs = synth.generate_config("cfg1_720p", n_frames=9, stream=60 + s, width=w, height=h)
        f = tempfile.NamedTemporaryFile(suffix=".ts", delete=False)
        f.write(synth.mux_ts(es, offs).tobytes())
        f.close()
        paths.append(f.name)
        want.append(cabi.decode_stream(oracle, es)[0])
    try:
        out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "hip_live_router.js")] + (["--devices", devices] if devices else []) + paths, timeout=300))
    finally:
        for p in paths:
            os.unlink(p)
    if devices:
        # {devices: [0, 0]}: the streams alternate between the two entries (least loaded first), every handle's pass is on the device
        # before the first is waited for -- two live handles in flight at once
        load = {"#0": 0, "#1": 0}
        for key, n in out["perHandle"]:
            load[key[-2:]] += n
        assert sorted(load.values()) == [2, 3] and out["waiting"] == 0 and {k.split("#")[0] for k, _ in out["perHandle"]} == {"%dx%d" % wh for wh in sizes}
    else:
        assert out["handles"] == sorted("%dx%d" % wh for wh in set(sizes)) and out["waiting"] == 0
    assert out["pictures"] == 5 * 9 and out["widths"] == [w for w, _ in sizes]
    for s, (w, h) in enumerate(sizes):
        st = out["streams"][s]
        assert st["planes"] == want[s], s
        assert st["sizes"] == [[w, h]] and st["frames"] == [[w, h]] * 9
