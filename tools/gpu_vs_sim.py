"""Diagnostic: decode one golden case on the GPU (batch engine) and in the
test-only simulator, then compare the intermediate tables stage by stage."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import batch as jb, synth  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "cfg0_240p_intra"
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "frames_%s.json" % case)))
es, offs = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
w, h = fx["info"]["width"], fx["info"]["height"]

simso = os.path.join(ROOT, "tests", "sim", "libjsmpeg_sim.so")
subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I",
                       os.path.join(ROOT, "jsmpeg_amd", "csrc"), "-o", simso,
                       os.path.join(ROOT, "tests", "sim", "sim_decode.cpp")])
sim = ctypes.CDLL(simso)
sim.sim_dump_counts.restype = ctypes.POINTER(ctypes.c_uint32)
cap = len(es) // 4 + 64
mbn = ((w + 15) // 16) * ((h + 15) // 16)
s_pos = np.zeros(cap, np.uint32); s_code = np.zeros(cap, np.uint8); s_own = np.zeros(cap, np.uint32)
s_mb = np.zeros((fx["n_frames"] + 2) * mbn * 16, np.uint8); s_tok = np.zeros((len(es) + 256) * 4, np.uint16)
sim.sim_set_dumps(ctypes.c_void_p(s_pos.ctypes.data), ctypes.c_void_p(s_code.ctypes.data),
                  ctypes.c_void_p(s_own.ctypes.data), ctypes.c_void_p(s_mb.ctypes.data), ctypes.c_void_p(s_tok.ctypes.data))
fb = fx["info"]["coded_size"] * 3 // 2
out = np.zeros(fx["n_frames"] * fb, np.uint8)
n = sim.sim_decode_stream(ctypes.c_void_p(es.ctypes.data), len(es), w, h, ctypes.c_void_p(out.ctypes.data), fx["n_frames"])
cnt = sim.sim_dump_counts()
n_sc, n_pics, _, ntok = cnt[0], cnt[1], cnt[2], cnt[3]
print("sim: frames", n, "n_sc", n_sc, "n_pics", n_pics)

b = jb.Batch(w, h, 1, fx["n_frames"] + 2, len(es) + 1024)
b.upload([es]); b.decode()
c = b.counters(); print("gpu:", c)
L = jb.lib()
L.jsmpeg_hip_batch_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
def rd(what, dtype, count):
    a = np.zeros(count, dtype)
    assert L.jsmpeg_hip_batch_debug_read(b.h, what, a.ctypes.data, 0, a.nbytes) == 0, jb.last_error()
    return a
g_pos = rd(0, np.uint32, n_sc); g_code = rd(1, np.uint8, n_sc); g_own = rd(2, np.uint32, n_sc)
for name, a, bb in (("sc_pos", g_pos, s_pos[:n_sc]), ("sc_code", g_code, s_code[:n_sc]), ("owner", g_own, s_own[:n_sc])):
    d = np.nonzero(a != bb)[0]
    print(name, "mismatches", len(d), d[:10], a[d[:5]], bb[d[:5]])
g_mb = rd(4, np.uint8, n_pics * mbn * 16).reshape(-1, 16); sm = s_mb[:n_pics * mbn * 16].reshape(-1, 16)
d = np.nonzero((g_mb != sm).any(axis=1))[0]
print("mbrec mismatches", len(d), d[:10], "pic", d[:10] // mbn, "addr", d[:10] % mbn)
for i in d[:4]:
    print("  gpu", g_mb[i], "\n  sim", sm[i])
# which slices lost macroblocks: first missing MB per (pic, row)
bad = {}
for i in d:
    bad.setdefault((int(i // mbn), int((i % mbn) // ((w + 15) // 16))), int(i % mbn))
rows = sorted(bad)
sl = np.nonzero(s_own[:n_sc] != 0xffffffff)[0]
key = {(int(s_own[i]), int(s_code[i]) - 1): int(i) for i in sl}
print("failing slices (sc index, block@256, lane, pos, first missing mb):")
for r in rows[:40]:
    i = key.get(r, -1)
    print("   ", i, i // 256, i % 256, int(s_pos[i]) if i >= 0 else -1, bad[r])
print("n failing slices", len(rows), "of", len(sl))
if int(os.environ.get("JSMPEG_HIP_DEBUG", "0")) & 4:
    dbg = rd(8, np.uint32, n_sc * 4).reshape(-1, 4)
    g_es = rd(7, np.uint8, 16 + len(es) + 64)
    print("device ES equals host ES:", bool((g_es[16:16 + len(es)] == es).all()))
    for r in rows[:12]:
        i = key.get(r, -1)
        reason, consumed, whi, wlo = [int(x) for x in dbg[i]]
        pos = int(s_pos[i]) + 4
        byte = pos + consumed // 8
        actual = bytes(g_es[byte:byte + 12]).hex()
        print("    sc", i, "reason", hex(reason), "consumed", consumed, "window %08x%08x" % (whi, wlo), "bitoff", consumed % 8,
              "ES@", byte, actual)
    ok = [int(i) for i in sl if (int(s_own[i]), int(s_code[i]) - 1) not in bad][:3]
    for i in ok:
        print("    ok sc", i, [hex(int(x)) for x in dbg[i]])
g_tok = rd(5, np.uint16, ntok)
# only compare slots the sim wrote (nonzero) -- the gpu buffer is not cleared
nz = np.nonzero(s_tok[:ntok])[0]
dt = nz[g_tok[nz] != s_tok[nz]]
print("token mismatches", len(dt), dt[:10])
hashes_ok = 0
import hashlib
for p in range(n):
    y, cr, cb = b.read_frame(p)
    got = hashlib.md5(y.tobytes() + cr.tobytes() + cb.tobytes()).hexdigest()
    simh = hashlib.md5(out[p * fb:(p + 1) * fb].tobytes()).hexdigest()
    if got != simh:
        fr = np.concatenate([y, cr, cb]); dd = np.nonzero(fr != out[p * fb:(p + 1) * fb])[0]
        print("frame", p, "differs at", len(dd), "bytes; first", dd[:8], "gpu", fr[dd[:8]], "sim", out[p * fb:(p + 1) * fb][dd[:8]])
    else:
        hashes_ok += 1
print("frames equal:", hashes_ok, "of", n)
