"""Regenerates tests/golden/frames_*.json (container only: needs /root/reference).

For each case the synthetic stream is produced by the committed generator
(jsmpeg_amd/csrc/synth_es.c, deterministic in its parameters), muxed into TS and
decoded by FOUR independent runs of the reference algorithm:
  1. reference src/mpeg1.js under Node              (oracle/ref_node_decode.js js)
  2. reference wasm build inlined in jsmpeg.min.js  (oracle/ref_node_decode.js wasm)
  3. reference src/wasm/*.c compiled natively       (oracle/_ref/libjsmpeg_ref.so)
  4. this repo's restatement                        (oracle/libmpeg1_oracle.so)
The fixture is written only if all four agree on every frame.  It stores the
generator parameters, md5 of the ES, and md5(Y|Cr|Cb) per frame, so the GPU box
(which has no /root/reference) can regenerate the input and check the output.

    python tests/golden/make_golden.py
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

CASES = {
    # name: (config, n_frames, overrides)
    "cfg0_240p_intra": ("cfg0_240p_intra", 30, {}),
    "cfg1_720p": ("cfg1_720p", 26, {}),
    "cfg2_1080p": ("cfg2_1080p", 25, {}),
    "cfg4_2160p": ("cfg4_2160p", 5, {}),
    "custom_quant_escapes": ("cfg1_720p", 14, dict(width=352, height=288, custom_quant=1, escape_permille=200,
                                                   dc_size_max=8, ac_max=12)),
    "quirk_levels": ("cfg1_720p", 14, dict(width=176, height=144, quirk_levels=1, escape_permille=300, ac_max=20)),
    "odd_size_17x33": ("cfg1_720p", 14, dict(width=17, height=33)),
    "wide_2048x64": ("cfg1_720p", 8, dict(width=2048, height=64, f_code_max=3)),
    "dense_high_rate": ("cfg1_720p", 7, dict(width=640, height=368, ac_max=40, coded_permille=950, qscale_lo=1,
                                             qscale_hi=31, dc_size_max=8)),
    "long_gop_p_chain": ("cfg1_720p", 40, dict(width=320, height=192, gop=40)),
    # valid but unusual syntax: slices starting / ending mid-row and spanning rows (first increments > 1, > 33 with
    # macroblock_escape on the wide one), extra_information_slice / _picture, macroblock_stuffing, extension and
    # user_data after the picture header
    "syntax_quirks_352x288": ("cfg1_720p", 14, dict(width=352, height=288, syntax_quirks=1)),
    "syntax_quirks_1280x96": ("cfg1_720p", 14, dict(width=1280, height=96, syntax_quirks=1, f_code_max=2)),
    # found by tools/fuzz_parity.py: the last macroblock of a slice is 6 bits long and sits in the slack of the slice's
    # last byte -- next_bytes_are_start_code (buffer.c:140-150) is already true after the macroblock before it, the
    # reference never decodes it and the picture shows what its plane set held two pictures earlier (mpeg1.c:1018-1020)
    "uncovered_last_mb_118x197": ("cfg1_720p", 18, dict(width=118, height=197, gop=14, ac_max=1, qscale_lo=7, qscale_hi=8,
                                                        escape_permille=0, custom_quant=1, quirk_levels=1, dc_size_max=2,
                                                        coded_permille=50, f_code_max=2, stream=2000)),
    # the same in the first P picture of a chain (picture 4 of I P P I P P ...): what it keeps showing belongs to the chain
    # before -- the batch engine's second reconstruct pass
    "uncovered_first_p_118x197": ("cfg1_720p", 12, dict(width=118, height=197, gop=3, ac_max=1, qscale_lo=7, qscale_hi=8,
                                                        escape_permille=0, custom_quant=0, quirk_levels=0, dc_size_max=2,
                                                        coded_permille=50, f_code_max=1, stream=3032)),
    # B and D pictures and P pictures with forward_f_code 0 between the decoded ones: consumed, not decoded, the plane sets
    # do not rotate (mpeg1.c:955-972).  decode() returns true for them too: the C ABI shows the previous picture again, the
    # reference's JS class does not render -- the four runs are compared without those repeats, and the C ABI's full list
    # is kept beside (abi_frame_md5)
    "skipped_pictures_352x288": ("cfg1_720p", 20, dict(width=352, height=288, syntax_quirks=2)),
    "long_slices_352x288": ("cfg1_720p", 16, dict(width=352, height=288, syntax_quirks=5)),
    "skipped_pictures_quirks_176x144": ("cfg1_720p", 20, dict(width=176, height=144, syntax_quirks=3, gop=5)),
    # coherent motion (one vector per picture + jitter): last macroblocks that only repeat the vector are 6 bits and get
    # lost in the slack of their slice's last byte in pictures 9, 11, 13, 17, 23 of I P I P ..., each the first P of its chain -- the
    # batch engine then lays its levels out across the GOPs (a picture after the frame its unwritten macroblocks show)
    "coherent_pan_352x288": ("cfg1_720p", 24, dict(width=352, height=288, gop=2, mv_jitter=1, f_code_max=1, coded_permille=60,
                                                   ac_max=1, stream=6045)),
}


# Valid syntax the generator never writes, added as byte stuffing on top of a generated stream (synth.stuff_zero_bytes):
# zero_byte stuffing in front of start codes.  A case is tried like every other; when the four runs agree it becomes a
# fixture (frames_*.json), when they do not the DISAGREEMENT is the record (excluded_*.json: which runs differ, on
# which pictures) -- the evidence behind "outside the contract" in DESIGN.md section 2.
PROBES = {
    # what a CBR encoder pads a picture with: 1..3 zero bytes between the last slice and the next picture start code
    "picture_end_stuffing_352x288": ("cfg1_720p", 14, dict(width=352, height=288, stuff_pictures=3)),
    # the same in front of slice start codes (next_start_code() inside a picture)
    "slice_stuffing_352x288": ("cfg1_720p", 14, dict(width=352, height=288, stuff_slices=3)),
}
CASES.update(PROBES)
# Streams from the independent encoder (tests/enc/mpeg1_enc.py: real motion search, DCT, quantisation, skipped / not-coded /
# intra decisions on procedural pictures), committed as tests/golden/enc_*.m1v: the same four-way agreement, the same fixture.
sys.path.insert(0, os.path.join(ROOT, "tests", "enc"))
import mpeg1_enc  # noqa: E402
for _name, _kw in mpeg1_enc.CASES.items():
    CASES["" + _name] = ("enc:" + _name, _kw["n_frames"], {})


def disagreement_record(name, cfg, n, ov, es, runs, why):
    """excluded_<name>.json: the four runs' hash lists compared picture by picture."""
    keys = sorted(runs)
    longest = max(len(v) for v in runs.values())
    rows = []
    for i in range(longest):
        hs = {k: (runs[k][i] if i < len(runs[k]) else None) for k in keys}
        if len(set(hs.values())) > 1:
            groups = {}
            for k, h in hs.items():
                groups.setdefault(h, []).append(k)
            rows.append({"picture": i, "groups": sorted(groups.values())})
    rec = dict(case=name, config=cfg, n_frames=n, overrides=ov, es_bytes=int(len(es)),
               es_md5=hashlib.md5(es.tobytes()).hexdigest(), verdict=why,
               decoded_frames={k: len(v) for k, v in runs.items()},
               pairs_equal={"%s==%s" % (a, b): runs[a] == runs[b] for i, a in enumerate(keys) for b in keys[i + 1:]},
               differing_pictures=len(rows), first_differences=rows[:16],
               frame_md5_by_run={k: v for k, v in runs.items()})
    with open(os.path.join(HERE, "excluded_%s.json" % name), "w") as fo:
        json.dump(rec, fo, indent=1)
    print("%-22s EXCLUDED: %s (%d pictures differ between runs)" % (name, why, len(rows)))


def node_hashes(ts_path, impl):
    out = subprocess.check_output(["node", os.path.join(ROOT, "oracle", "ref_node_decode.js"), ts_path, impl])
    return json.loads(out)["hashes"]


def main():
    build.build_synth(); build.build_oracle(); build.build_ref()
    only = sys.argv[1:]
    for name, (cfg, n, ov) in CASES.items():
        if only and name not in only:
            continue
        es, offs = synth.generate_config(cfg, n_frames=n, **ov)
        ts = synth.mux_ts(es, offs)
        with tempfile.NamedTemporaryFile(suffix=".ts", delete=False) as f:
            f.write(ts.tobytes())
        try:
            runs = {
                "ref_js": node_hashes(f.name, "js"),
                "ref_wasm": node_hashes(f.name, "wasm"),
                "ref_native": cabi.decode_stream(build.LIB_REF, es, offs)[0],
                "oracle": cabi.decode_stream(build.LIB_ORACLE, es, offs)[0],
            }
        finally:
            os.unlink(f.name)
        abi_list = runs["ref_native"]
        if runs["oracle"] != abi_list and name not in PROBES:
            raise SystemExit("%s: the restatement disagrees with the reference's C on the full decode() sequence" % name)
        if name in PROBES:
            wr = {k: [h for i, h in enumerate(v) if i == 0 or h != v[i - 1]] for k, v in runs.items()}
            if any(v != wr["ref_js"] for v in wr.values()) or len(wr["ref_js"]) != n:
                why = ("the reference's own JS and wasm / C decoders disagree" if wr["ref_js"] != wr["ref_wasm"] or wr["ref_js"] != wr["ref_native"]
                       else "the reference decodes %d of %d pictures" % (len(wr["ref_js"]), n))
                disagreement_record(name, cfg, n, ov, es, runs, why)
                continue
            stale = os.path.join(HERE, "excluded_%s.json" % name)
            if os.path.exists(stale):
                os.unlink(stale)

        def without_repeats(v):
            return [h for i, h in enumerate(v) if i == 0 or h != v[i - 1]]

        runs = {k: without_repeats(v) for k, v in runs.items()}
        first = runs["ref_js"]
        for k, v in runs.items():
            if v != first:
                raise SystemExit("%s: %s disagrees with ref_js - not writing a fixture" % (name, k))
        if len(first) != n:
            raise SystemExit("%s: decoded %d of %d frames" % (name, len(first), n))
        _, idx, info = cabi.decode_stream(build.LIB_REF, es)
        fixture = dict(case=name, config=cfg, n_frames=n, overrides=ov, es_bytes=int(len(es)),
                       es_md5=hashlib.md5(es.tobytes()).hexdigest(), agreed_by=sorted(runs), info=info,
                       bit_index_after_decode=idx, frame_md5=first)
        if abi_list != first:
            fixture["abi_frame_md5"] = abi_list   # one entry per decode() == true, skipped pictures repeat the previous one
        with open(os.path.join(HERE, "frames_%s.json" % name), "w") as fo:
            json.dump(fixture, fo, indent=1)
        print("%-22s %3d frames  %9d ES bytes  all four agree" % (name, n, len(es)))


if __name__ == "__main__":
    main()
