"""Regenerates tests/golden/rgba_*.json (container only: needs /root/reference).

Renderer stage.  For each case the synthetic stream (committed generator) is muxed into TS and
  1. decoded by the reference's src/mpeg1.js connected to the reference's src/canvas2d.js over a stub canvas under
     Node (oracle/ref_node_rgba.js): md5 of imageData.data after every frame;
  2. decoded by this repo's decoder restatement and converted by oracle/ycbcr_oracle.c.
The fixture (generator parameters + md5 per frame) is written only if both agree on every frame.

    python tests/golden/make_golden_rgba.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from jsmpeg_amd import build, cabi, synth  # noqa: E402
from oracle import checkers

CASES = {
    "cfg0_240p_intra": ("cfg0_240p_intra", 6, {}),
    "cif_352x288": ("cfg1_720p", 13, dict(width=352, height=288)),
    "odd_size_17x33": ("cfg1_720p", 8, dict(width=17, height=33)),
    "width_not_multiple_of_4_150x98": ("cfg1_720p", 8, dict(width=150, height=98)),
    "cfg2_1080p": ("cfg2_1080p", 3, {}),
}


def main():
    build.build_synth(); build.build_oracle()
    for name, (cfg, n, ov) in CASES.items():
        es, offs = synth.generate_config(cfg, n_frames=n, **ov)
        ts = synth.mux_ts(es, offs)
        with tempfile.NamedTemporaryFile(suffix=".ts", delete=False) as f:
            f.write(ts.tobytes())
        try:
            ref = json.loads(subprocess.check_output(["node", os.path.join(ROOT, "oracle", "ref_node_rgba.js"), f.name]))
        finally:
            os.unlink(f.name)
        frames, _, info = cabi.decode_stream(build.LIB_ORACLE, es, keep="planes")
        mine = [hashlib.md5(checkers.oracle_rgba(build.LIB_ORACLE, y, cr, cb, info["width"], info["height"]).tobytes()).hexdigest()
                for y, cr, cb in frames]
        assert ref["frames"] == n == len(mine), (name, ref["frames"], len(mine))
        assert (ref["width"], ref["height"]) == (info["width"], info["height"])
        assert ref["hashes"] == mine, "%s: reference canvas2d.js and oracle/ycbcr_oracle.c disagree" % name
        out = dict(config=cfg, n_frames=n, overrides=ov, width=info["width"], height=info["height"],
                   es_md5=hashlib.md5(es.tobytes()).hexdigest(), rgba_md5=mine,
                   agreed_by=["reference src/mpeg1.js + src/canvas2d.js under Node", "oracle/mpeg1_oracle.c + oracle/ycbcr_oracle.c"])
        with open(os.path.join(HERE, "rgba_%s.json" % name), "w") as fh:
            json.dump(out, fh, indent=1)
        print(name, "ok:", n, "frames", info["width"], "x", info["height"])


if __name__ == "__main__":
    main()
