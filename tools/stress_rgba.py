"""Stress: decode + render_rgba through the one-picture ABI, repeatedly; reports which check fails."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import build, cabi, synth
from oracle import checkers
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
es, offs = synth.generate_config("cfg2_1080p", n_frames=3)
frames, _, info = cabi.decode_stream(build.LIB_ORACLE, es, keep="planes")
want = [checkers.oracle_rgba(build.LIB_ORACLE, y, cr, cb, 1920, 1080) for y, cr, cb in frames]
bad = 0
for r in range(reps):
    with cabi.Mpeg1Decoder(build.LIB_HIP, len(es) + 1024, cabi.MODE_EXPAND) as dec:
        dec.write(es)
        k = 0
        while dec.decode():
            rgba = dec.render_rgba()
            pl = dec.planes()
            ok_planes = all(np.array_equal(a, b) for a, b in zip(pl, frames[k]))
            ok_rgba = np.array_equal(rgba, want[k])
            if not (ok_planes and ok_rgba):
                bad += 1
                d = np.argwhere((rgba != want[k]).any(axis=2))
                print("rep %d frame %d: planes %s rgba %s; %d differing pixels, first %s last %s" % (
                    r, k, ok_planes, ok_rgba, len(d), d[:1].tolist(), d[-1:].tolist()))
            k += 1
        assert k == 3
print("bad", bad, "of", reps * 3)
