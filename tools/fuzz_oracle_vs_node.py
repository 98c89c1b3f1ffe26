"""CPU, build container only (needs /root/reference and node): random streams from the generator's whole parameter space,
muxed to MPEG-TS, decoded by the UNMODIFIED reference under Node -- src/mpeg1.js and the shipped wasm build, through the
reference's own src/ts.js (oracle/ref_node_decode.js) -- and by the restatement (oracle/libmpeg1_oracle.so): every picture's
md5(Y|Cr|Cb) must agree three ways.  The pin of the checker beyond the committed fixtures.
    python tools/fuzz_oracle_vs_node.py [cases] [seed]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import build, cabi, synth  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    w, h = int(rng.integers(1, 30)) * 16 - int(rng.integers(0, 16)), int(rng.integers(1, 20)) * 16 - int(rng.integers(0, 16))
    ov = dict(width=max(w, 2), height=max(h, 2), gop=int(rng.choice([1, 2, 5, 12, 40])), ac_max=int(rng.choice([0, 3, 24, 63])),
              coded_permille=int(rng.choice([50, 400, 950])), f_code_max=int(rng.integers(1, 8)), syntax_quirks=int(rng.choice([0, 0, 1, 2, 3, 5, 7])),
              mv_jitter=int(rng.choice([0, 2, 6])), qscale_lo=int(rng.integers(1, 8)), qscale_hi=int(rng.integers(8, 32)),
              escape_permille=int(rng.choice([0, 20, 300, 1000])), custom_quant=int(rng.integers(0, 2)), quirk_levels=int(rng.integers(0, 2)),
              dc_size_max=int(rng.integers(0, 9)))
    if ov["syntax_quirks"] & 2:
        ov["ac_max"], ov["dc_size_max"] = max(ov["ac_max"], 1), max(ov["dc_size_max"], 2)
    n = int(rng.integers(2, 16))
    try:
        es, offs = synth.generate_config("cfg1_720p", n_frames=n, stream=90000 + c, **ov)
    except RuntimeError as e:
        print("case %d: generator: %s" % (c, e)); continue
    want, _, _ = cabi.decode_stream(build.LIB_ORACLE, es, offs)
    ts = synth.mux_ts(es, offs)
    with tempfile.NamedTemporaryFile(suffix=".ts", delete=False) as f:
        f.write(ts.tobytes())
    try:
        got = {impl: json.loads(subprocess.check_output(["node", os.path.join(ROOT, "oracle", "ref_node_decode.js"), f.name, impl]))["hashes"]
               for impl in ("js", "wasm")}
    finally:
        os.unlink(f.name)
    # a picture that is consumed without being decoded (B / D / forward_f_code 0: syntax_quirks & 2) makes decode() return true with
    # the planes of the picture before: the wasm class renders them again, the JS class does not render, the restatement's loop
    # sees them again -- compared: the sequence of DISTINCT pictures (the generator keeps consecutive real pictures different then)
    def distinct(xs):
        return [x for i, x in enumerate(xs) if i == 0 or x != xs[i - 1]]
    want, got = distinct(want), {k: distinct(v) for k, v in got.items()}
    # ts.js hands a picture on when a packet WITH stuffing ends it or the next picture begins (ts.js:143-146): a last picture whose
    # last packet is exactly full never leaves the reference's demuxer (1 stream in 184) -- then the reference has one picture fewer
    if not (int(ts[-188 + 3]) & 0x20) and all(len(v) == len(want) - 1 for v in got.values()):
        want = want[:-1]
    if got["js"] != want or got["wasm"] != want:
        bad += 1
        print("case %d MISMATCH: js %s wasm %s | n %d %r" % (c, [a == b for a, b in zip(got["js"], want)], [a == b for a, b in zip(got["wasm"], want)], n, ov), flush=True)
print("%d cases, %d mismatches (reference JS and wasm under Node against the restatement)" % (cases, bad))
sys.exit(1 if bad else 0)
