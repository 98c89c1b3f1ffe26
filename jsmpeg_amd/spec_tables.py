"""Parses jsmpeg_amd/csrc/mpeg1_vlc_codes.h (the product's MPEG-1 VLC/constant
tables) into Python dicts so tests and tools can use the exact same data the
HIP code and the synthetic generator are compiled from."""
import os
import re

_HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "mpeg1_vlc_codes.h")


def _macro_body(text, name):
    m = re.search(r"#define\s+%s(?:\(X\))?\s+((?:.*\\\n)*.*)\n" % re.escape(name), text)
    if not m:
        raise KeyError(name)
    return m.group(1).replace("\\\n", " ")


def load(path=None):
    text = open(path or _HDR).read()
    out = {}
    for name in ("MBA", "MBTYPE_I", "MBTYPE_P", "CBP", "MOTION", "DCSIZE_LUMA", "DCSIZE_CHROMA"):
        body = _macro_body(text, "MPEG1_VLC_" + name)
        out[name] = {b: int(v, 0) for b, v in re.findall(r'X\("([01]+)",\s*(-?\w+)\)', body)}
    body = _macro_body(text, "MPEG1_VLC_DCT_COEFF")
    out["DCT_COEFF"] = {b: (int(r), int(l)) for b, r, l in re.findall(r'X\("([01]+)",\s*(\d+),\s*(\d+)\)', body)}
    out["DCT_ESCAPE"] = re.search(r'MPEG1_VLC_DCT_ESCAPE_BITS\s+"([01]+)"', text).group(1)
    for name in ("ZIGZAG", "DEFAULT_INTRA_QUANT", "PREMULTIPLIER"):
        body = _macro_body(text, "MPEG1_%s_INIT" % name)
        out[name] = [int(x) for x in re.findall(r"\d+", body)]
    body = _macro_body(text, "MPEG1_PICTURE_RATE_INIT")
    out["PICTURE_RATE"] = [float(x) for x in re.findall(r"\d+\.\d+", body)]
    return out
