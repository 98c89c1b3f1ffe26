"""Container only (needs /root/reference through oracle/_ref and Node): the golden vectors of tests/golden/enc1080/ --
four 1080p GOPs of 12 pictures from the test-side encoder (tools/enc_content.py writes them from tests/enc/mpeg1_enc.py:
GOPs 0, 2, 4, 6 of its table, ~16 Mbit/s together: the content of bench.py's `coded_video_content` and of
tools/enc_content_bench.py).  A fixture is written only when the reference's JS decoder, its wasm build (both under Node,
fed the stream as MPEG-TS through the reference's own ts.js), its C sources compiled natively and the restatement in
oracle/ agree on every picture -- the rule of make_golden.py, whose helpers this uses.
    python tests/golden/make_golden_enc1080.py"""
import glob
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden  # noqa: E402  (node_hashes)
from jsmpeg_amd import build, cabi, synth  # noqa: E402


def picture_offsets(es):
    """what the encoder returned beside the stream: where each picture's bytes begin (the first picture's range begins with the
    sequence header), and the sequence end code behind the last"""
    at = np.flatnonzero((es[:-3] == 0) & (es[1:-2] == 0) & (es[2:-1] == 1) & (es[3:] == 0))
    return np.concatenate([[0], at[1:], [len(es) - 4]]).astype(np.uint32)


def main():
    build.build_synth(); build.build_oracle(); build.build_ref()
    for path in sorted(glob.glob(os.path.join(HERE, "enc1080", "enc1080_*.m1v"))):
        name = os.path.basename(path)[:-4]
        es = np.fromfile(path, dtype=np.uint8)
        offs = picture_offsets(es)
        ts = synth.mux_ts(es, offs)
        with tempfile.NamedTemporaryFile(suffix=".ts", delete=False) as f:
            f.write(ts.tobytes())
        try:
            runs = {"ref_js": make_golden.node_hashes(f.name, "js"), "ref_wasm": make_golden.node_hashes(f.name, "wasm"),
                    "ref_native": cabi.decode_stream(build.LIB_REF, es, offs)[0], "oracle": cabi.decode_stream(build.LIB_ORACLE, es, offs)[0]}
        finally:
            os.unlink(f.name)
        first = runs["ref_js"]
        for k, v in runs.items():
            if v != first:
                raise SystemExit("%s: %s disagrees with ref_js - not writing a fixture" % (name, k))
        if len(first) != len(offs) - 1:
            raise SystemExit("%s: decoded %d of %d pictures" % (name, len(first), len(offs) - 1))
        fixture = dict(case=name, made_by="tools/enc_content.py (tests/enc/mpeg1_enc.py)", n_frames=len(first), es_bytes=int(len(es)),
                       es_md5=hashlib.md5(es.tobytes()).hexdigest(), agreed_by=sorted(runs), frame_md5=first)
        with open(os.path.join(HERE, "enc1080", "frames_%s.json" % name), "w") as fo:
            json.dump(fixture, fo, indent=1)
        print("%-12s %3d pictures  %9d ES bytes  all four agree" % (name, len(first), len(es)))


if __name__ == "__main__":
    main()
