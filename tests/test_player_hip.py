"""JSMpeg.PlayerHIP (jsmpeg_amd/js/player-hip.js): the reference's own Player with its decoder selection
(src/player.js:35-38, 48-52) resolved to the HIP classes without editing player.js.  Container only: player.js is the
reference's file and does not travel to the GPU box; the classes themselves are covered there by the other Node tests."""
import json
import os
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT, have_reference

NODE = shutil.which("node")
pytestmark = [pytest.mark.skipif(NODE is None, reason="node not installed"), pytest.mark.reference,
              pytest.mark.skipif(not have_reference(), reason="needs /root/reference")]


@pytest.mark.parametrize("mode", ["static", "streaming"])
def test_player_with_hip_decoders_behaves_like_the_reference_player(mode):
    """An A/V transport stream played by JSMpeg.Player (its wasm decoders) and by JSMpeg.PlayerHIP (native binding =
    stand-ins over the same wasm exports) under recording DOM stand-ins with a test-driven clock: every rendered
    frame (md5 of the Canvas2D RGBA), every audio buffer (md5, start time), currentTime at every animation frame and
    the seek must be identical; the HIP classes must have been the ones constructed, and the names restored."""
    from test_mp2_gpu import _av_ts
    ts, es, fx, data = _av_ts(30, "stereo_44k_192", 3)
    f = tempfile.NamedTemporaryFile(suffix=".ts", delete=False)
    f.write(ts.tobytes())
    f.close()
    try:
        args = [NODE, os.path.join(ROOT, "tests", "js", "player_vs_reference.js"), f.name]
        if mode == "streaming":
            args.append("streaming")
        out = json.loads(subprocess.check_output(args))
    finally:
        os.unlink(f.name)
    assert out["same"], out
    assert out["hipClassesSelected"]
    assert out["frames"] >= 15 and out["audio"] >= 10


def test_install_needs_the_player():
    script = ("try{require(%r).install({});console.log('NO THROW')}catch(e){console.log('THROWS:'+e.message)}"
              % os.path.join(ROOT, "jsmpeg_amd", "js", "player-hip.js"))
    assert subprocess.check_output([NODE, "-e", script]).decode().startswith("THROWS:")
