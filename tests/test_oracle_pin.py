"""Pins the oracle (oracle/mpeg1_oracle.c): against the committed golden
fixtures everywhere, and against live runs of the reference (native C, JS and
wasm under Node) where /root/reference exists."""
import glob
import hashlib
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT, have_reference
from jsmpeg_amd import cabi, synth

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "frames_*.json")))


def load_case(path):
    fx = json.load(open(path))
    es, offs = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
    assert hashlib.md5(es.tobytes()).hexdigest() == fx["es_md5"], "generator drifted from the fixture"
    return fx, es, offs


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[7:-5] for p in FIXTURES])
def test_oracle_matches_golden(path, libs):
    fx, es, offs = load_case(path)
    frames, idx, info = cabi.decode_stream(libs["oracle"], es)
    # one entry per decode() == true; pictures the reference consumes without decoding (B / D / f_code 0) repeat the previous one
    assert frames == fx.get("abi_frame_md5", fx["frame_md5"])
    assert idx == fx["bit_index_after_decode"]
    assert info["coded_size"] == fx["info"]["coded_size"] and info["width"] == fx["info"]["width"]
    # streaming-style feed (one write per picture, ts.js:205-210) must give the same pictures
    frames2, _, _ = cabi.decode_stream(libs["oracle"], es, offs)
    assert frames2 == fx.get("abi_frame_md5", fx["frame_md5"])


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[7:-5] for p in FIXTURES])
def test_reference_native_matches_golden(path, libs):
    if not libs["ref"] or not os.path.exists(libs["ref"]):
        pytest.skip("oracle/_ref not built (needs /root/reference once)")
    fx, es, offs = load_case(path)
    frames, idx, _ = cabi.decode_stream(libs["ref"], es)
    assert frames == fx.get("abi_frame_md5", fx["frame_md5"])
    assert idx == fx["bit_index_after_decode"]


def test_oracle_evict_mode_streaming(libs):
    """EVICT (streaming) store: small buffer, write a picture / decode a picture
    (buffer.c:167-190).  Same pictures as the one-shot EXPAND decode."""
    fx, es, offs = load_case(os.path.join(ROOT, "tests", "golden", "frames_cfg0_240p_intra.json"))
    for lib in [libs["oracle"]] + ([libs["ref"]] if libs["ref"] and os.path.exists(libs["ref"]) else []):
        got = []
        with cabi.Mpeg1Decoder(lib, 24 * 1024, cabi.MODE_EVICT) as dec:
            n = len(offs) - 1
            for k in range(n):
                end = len(es) if k == n - 1 else int(offs[k + 1])
                dec.write(es[int(offs[k]):end])
                while dec.decode():
                    h = hashlib.md5()
                    for p in dec.planes():
                        h.update(p.tobytes())
                    got.append(h.hexdigest())
        assert got == fx["frame_md5"], lib


def test_oracle_unit_idct_dc_only(libs):
    import ctypes
    lib = ctypes.CDLL(libs["oracle"])
    blk = (ctypes.c_int32 * 64)()
    for dc in (-2048 * 32, -1000, -129, -128, 0, 127, 128, 255 * 256, 2047 * 32):
        for i in range(64):
            blk[i] = 0
        blk[0] = dc
        lib.oracle_idct(blk)
        assert list(blk) == [(dc + 128) >> 8] * 64  # the reference's n == 1 shortcut, mpeg1.c:1578-1581


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference")
@pytest.mark.parametrize("impl", ["js", "wasm"])
def test_reference_under_node_matches_golden(impl, libs):
    fx, es, offs = load_case(os.path.join(ROOT, "tests", "golden", "frames_custom_quant_escapes.json"))
    ts = synth.mux_ts(es, offs)
    with tempfile.NamedTemporaryFile(suffix=".ts", delete=False) as f:
        f.write(ts.tobytes())
    try:
        out = json.loads(subprocess.check_output(["node", os.path.join(ROOT, "oracle", "ref_node_decode.js"),
                                                  f.name, impl]))
    finally:
        os.unlink(f.name)
    assert out["hashes"] == fx["frame_md5"]
    assert out["sizes"] == [[fx["info"]["width"], fx["info"]["height"]]]


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference and node")
def test_oracle_equals_the_reference_under_node_on_random_streams(libs):
    """tools/fuzz_oracle_vs_node.py, a short run: random streams of the generator's whole parameter space through the unmodified
    reference's JS and wasm decoders under Node (its own ts.js in front) and through the restatement: the same pictures."""
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_oracle_vs_node.py"), "30", "3"], capture_output=True, text=True)
    assert out.returncode == 0 and "30 cases, 0 mismatches" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_oracle_equals_the_references_c_build_on_random_call_patterns(libs):
    """tools/fuzz_abi_chunks.py with the reference's own C build (oracle/_ref) in the product's place: random chunking, EVICT
    stores that evict, EXPAND stores that grow, partial pulls, seeks -- every decode()'s return value, cursor and planes of the
    restatement against the reference's (buffer.c + mpeg1.c)."""
    import sys
    if not os.path.exists(libs.get("ref") or ""):
        pytest.skip("oracle/_ref/libjsmpeg_ref.so not built")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_abi_chunks.py"), "150", "8"], capture_output=True, text=True,
                         env=dict(os.environ, FUZZ_LIB=libs["ref"]))
    assert out.returncode == 0 and "150 cases, 0 mismatches" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="container only: pins the test-side encoder to the streams it committed")
def test_encoder_script_reproduces_the_committed_streams():
    """tests/golden/enc_*.m1v are what tests/enc/mpeg1_enc.py writes (float DCT: checked where the fixtures were made)."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "enc"))
    import mpeg1_enc
    name = "enc_coarse_fullpel_160x128"
    es, offs = mpeg1_enc.encode(**mpeg1_enc.CASES[name])
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "frames_%s.json" % name)))
    assert hashlib.md5(es.tobytes()).hexdigest() == fx["es_md5"] and len(offs) == fx["n_frames"] + 1


def _stream_with_invalid_cbp_and_dc_size():
    """32 x 16, two pictures written bit by bit: an I picture whose first luma block has the dct_dc_size code 1111111 (no
    such code), and a P picture whose first macroblock has the coded_block_pattern 00000000 (no code begins like that).
    The reference's tree walk returns its table's T[1] for both: size 3 and pattern 3 (mpeg1.c:202, 401, 1742-1748)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "enc"))
    import mpeg1_enc as E
    w = E.Bits()
    w.start_code(0xB3)
    w.put(32, 12); w.put(16, 12); w.put(1, 4); w.put(5, 4); w.put(0x3FFFF, 18); w.put(1, 1); w.put(20, 10); w.put(0, 1); w.put(0, 1); w.put(0, 1)
    w.start_code(0xB8)
    w.put(0, 1); w.put(0, 5); w.put(0, 6); w.put(1, 1); w.put(0, 6); w.put(0, 6); w.put(1, 1); w.put(0, 1)
    # ---- I picture ----
    w.start_code(0x00)
    w.put(0, 10); w.put(1, 3); w.put(0xFFFF, 16); w.put(0, 1)
    w.start_code(0x01)
    w.put(8, 5); w.put(0, 1)
    for mb in range(2):
        w.code(E.INV["MBA"][1]); w.code(E.INV["MBTYPE_I"][0x01])
        for b in range(6):
            if mb == 0 and b == 0:
                w.code("1111111")            # no dct_dc_size_luminance code: read as size 3
                w.put(0b101, 3)              # its three differential bits: +5
            else:
                w.code(E.INV["DCSIZE_LUMA" if b < 4 else "DCSIZE_CHROMA"][0])
            w.code("10")                     # end_of_block
    # ---- P picture ----
    w.start_code(0x00)
    w.put(1, 10); w.put(2, 3); w.put(0xFFFF, 16); w.put(0, 1); w.put(1, 3); w.put(0, 1)
    w.start_code(0x01)
    w.put(8, 5); w.put(0, 1)
    w.code(E.INV["MBA"][1]); w.code(E.INV["MBTYPE_P"][0x02])
    w.code("00000000")                       # no coded_block_pattern code: read as pattern 3 = blocks 4 and 5
    for b in range(2):
        w.code("1"); w.put(b, 1); w.code("10")   # (0, +1) / (0, -1), end_of_block
    # the second macroblock intra (31 bits: one of 6 bits would hide in the slice's last byte and never be decoded, mpeg1.c:1018-1020)
    w.code(E.INV["MBA"][1]); w.code(E.INV["MBTYPE_P"][0x01])
    for b in range(6):
        w.code(E.INV["DCSIZE_LUMA" if b < 4 else "DCSIZE_CHROMA"][0]); w.code("10")
    w.start_code(0xB7)
    return np.frombuffer(bytes(w.out), dtype=np.uint8).copy()


def test_invalid_cbp_and_dc_size_read_as_the_reference_reads_them(libs):
    """read_huffman returns T[1] for a bit string that is no code -- 3, not 6, in the three tables that begin
    2*3, 1*3, 0.  Only damaged streams get there; the restatement is held against the reference's own build all the same."""
    if not libs["ref"] or not os.path.exists(libs["ref"]):
        pytest.skip("oracle/_ref not built (needs /root/reference once)")
    es = _stream_with_invalid_cbp_and_dc_size()
    want, want_idx, _ = cabi.decode_stream(libs["ref"], es)
    got, got_idx, _ = cabi.decode_stream(libs["oracle"], es)
    assert len(want) == 2 and got == want and got_idx == want_idx
    # and the values really are in play: the +5 differential moved the first block's DC, blocks 4 / 5 of the P picture got their +-1
    frames, _, _ = cabi.decode_stream(libs["oracle"], es, keep="planes")
    (y0, cr0, cb0), (y1, cr1, cb1) = frames
    assert int(y0.reshape(16, 32)[0, 0]) == 128 + 5          # (the predictor carries it to the blocks behind)
    assert not np.array_equal(cr1.reshape(8, 16)[:, :8], cr0.reshape(8, 16)[:, :8]) and not np.array_equal(cb1.reshape(8, 16)[:, :8], cb0.reshape(8, 16)[:, :8])
