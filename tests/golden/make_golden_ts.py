"""Regenerates tests/golden/ts_*.json (container only: needs /root/reference).

Ingest side.  Each crafted TS (tests/ts_craft.py, deterministic) is demuxed by
  1. the reference's src/ts.js under Node (oracle/ref_node_ts.js), ONE write() of the whole buffer, stream 0xE0
     connected -- and, for the cases of ts_craft.WRITES, the same buffer in several write() calls (leftover bytes);
  2. the CPU restatement oracle/ts_oracle.c.
The fixture (md5 of the TS, and per destination.write call its pts, byte count and md5) is written only if both agree.

    python tests/golden/make_golden_ts.py
"""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ts_craft  # noqa: E402
from jsmpeg_amd import build, cabi  # noqa: E402
from oracle import checkers


def one(name, ts, write_sizes):
    with tempfile.NamedTemporaryFile(suffix=".ts", delete=False) as f:
        f.write(ts.tobytes())
    try:
        cmd = ["node", os.path.join(ROOT, "oracle", "ref_node_ts.js"), f.name, "224"]
        if write_sizes:
            cmd.append(",".join(str(x) for x in write_sizes))
        ref = json.loads(subprocess.check_output(cmd))
    finally:
        os.unlink(f.name)
    es, writes = checkers.oracle_ts_demux(build.LIB_ORACLE, ts, 0xE0, write_sizes)
    mine = [dict(pts=p, length=int(n), md5=hashlib.md5(es[o:o + n].tobytes()).hexdigest()) for p, o, n in writes]
    assert len(mine) == len(ref["writes"]), (name, len(mine), len(ref["writes"]))
    for a, b in zip(mine, ref["writes"]):
        assert a["length"] == b["length"] and a["md5"] == b["md5"] and a["pts"] == b["pts"], (name, a, b)
    out = dict(case=name, ts_md5=hashlib.md5(ts.tobytes()).hexdigest(), ts_bytes=int(len(ts)), stream_id=0xE0,
               writes=ref["writes"], total_md5=ref["total_md5"],
               agreed_by=["reference src/ts.js under Node", "oracle/ts_oracle.c"])
    fname = "ts_%s.json" % name
    if write_sizes:
        out["write_sizes"] = write_sizes
        fname = "ts_%s__in_%d_writes.json" % (name, len(write_sizes))
    with open(os.path.join(HERE, fname), "w") as fh:
        json.dump(out, fh, indent=1)
    print(fname, "ok:", len(mine), "writes,", sum(w["length"] for w in mine), "bytes")


def main():
    build.build_synth(); build.build_oracle(force=True)
    for name, fn in ts_craft.CASES.items():
        ts = fn()
        one(name, ts, None)
        if name in ts_craft.WRITES:
            one(name, ts, ts_craft.WRITES[name])


if __name__ == "__main__":
    main()
