"""The product's and the oracle's Annex-B tables vs the dump of the reference's
own VLC trees (tests/golden/vlc_codes.json, produced by oracle/dump_ref_vlc.js
from reference src/mpeg1.js:986-1663)."""
import json
import math
import os
import re
import subprocess

import pytest

from conftest import ROOT, have_reference
from jsmpeg_amd import spec_tables

GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "vlc_codes.json")))


def _check(t):
    for k in ("MBA", "MBTYPE_I", "MBTYPE_P", "CBP", "MOTION", "DCSIZE_LUMA", "DCSIZE_CHROMA"):
        assert t[k] == GOLDEN[k], k
    coeff = {b: (r << 8) | l for b, (r, l) in t["DCT_COEFF"].items()}
    coeff["1"] = 0x0001
    coeff[t["DCT_ESCAPE"]] = 0xFFFF
    assert coeff == GOLDEN["DCT_COEFF"]
    assert t["ZIGZAG"] == GOLDEN["ZIG_ZAG"]
    assert t["DEFAULT_INTRA_QUANT"] == GOLDEN["DEFAULT_INTRA_QUANT_MATRIX"]
    assert t["PREMULTIPLIER"] == GOLDEN["PREMULTIPLIER_MATRIX"]
    assert t["PICTURE_RATE"] == [float(x) for x in GOLDEN["PICTURE_RATE"]]
    assert GOLDEN["DEFAULT_NON_INTRA_QUANT_MATRIX"] == [16] * 64


def test_product_tables_match_reference_dump():
    _check(spec_tables.load())


def test_oracle_tables_match_reference_dump():
    _check(spec_tables.load(os.path.join(ROOT, "oracle", "annex_b_codes.h")))


def test_codes_are_prefix_free():
    t = spec_tables.load()
    for k in ("MBA", "MBTYPE_I", "MBTYPE_P", "CBP", "MOTION", "DCSIZE_LUMA", "DCSIZE_CHROMA"):
        codes = sorted(t[k])
        for a in codes:
            for b in codes:
                assert a == b or not b.startswith(a), (k, a, b)
    codes = sorted(list(t["DCT_COEFF"]) + ["1", t["DCT_ESCAPE"]])
    for a in codes:
        for b in codes:
            assert a == b or not b.startswith(a), (a, b)


def test_premultiplier_formula():
    """P[8i+j] = round(32 a_i a_j), a_0 = 1, a_k = sqrt(2) cos(k pi / 16)."""
    a = [1.0] + [math.sqrt(2.0) * math.cos(k * math.pi / 16.0) for k in range(1, 8)]
    want = [int(round(32.0 * a[i] * a[j])) for i in range(8) for j in range(8)]
    assert spec_tables.load()["PREMULTIPLIER"] == want


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference")
def test_golden_dump_is_current():
    out = subprocess.check_output(["node", os.path.join(ROOT, "oracle", "dump_ref_vlc.js")])
    assert json.loads(out) == GOLDEN
