"""CPU (minutes): 1080p GOPs from the test-side ENCODER (tests/enc/mpeg1_enc.py: procedural moving pictures, block motion
search + half-pel refinement, DCT, quantisation, skipped / not-coded macroblocks) for tools/enc_content_bench.py -- coded
VIDEO statistics (coherent vector fields, zero vectors, skipped runs, sparse high frequencies) at the headline's picture
size, beside the generator's uniform-random syntax.  Eight GOPs of 12 pictures with different motion, noise and quantiser,
one process each -> tests/enc/_cache/enc1080_<k>.m1v (python tools/enc_content.py [pictures per GOP] [GOP numbers, e.g. 8,9,10,11]) (git-ignored: 15-50 s of Python per picture; they travel to the GPU box
with the tree like the built libraries).  GOPs 0, 2, 4 and 6 (quantiser 6-10, the short search range: ~16 Mbit/s per stream, the
headline's bit rate) are COMMITTED as tests/golden/enc1080/ with golden vectors (tests/golden/make_golden_enc1080.py: reference
JS == wasm == C == oracle): bench.py's `coded_video_content` and tests/test_enc1080_golden.py read those.    python tools/enc_content.py [pictures per GOP]"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "enc"))
OUT = os.path.join(ROOT, "tests", "enc", "_cache")
GOPS = (  # seed, pan (pixels per picture), noise (sigma), quantiser scale, f_code
    (11, 1.5, 1.5, 8, 1), (12, 3.0, 2.5, 6, 2), (13, 0.0, 1.0, 8, 1), (14, 5.0, 3.0, 5, 2),
    (15, 0.7, 2.0, 10, 1), (16, 7.0, 4.0, 4, 3), (17, 2.0, 0.5, 6, 1), (18, 4.0, 6.0, 3, 2))
# (coarser quantisers do not make this content's intra pictures small: at quantiser 16 / 24 / 31 they are still 0.36 MB -- the
# procedural texture is a random walk; intra pictures BELOW the density at which the reconstruct's plan changes (19.4 bytes per
# macroblock) have only been measured on the generator's content)


def one(k):
    import mpeg1_enc as enc
    seed, pan, noise, q, f = GOPS[k]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    t0 = time.time()
    es, offs = enc.encode(width=1920, height=1080, n_frames=n, gop=n, qscale=q, f_code=f, seed=seed, pan=pan, noise=noise)
    es.tofile(os.path.join(OUT, "enc1080_%d.m1v" % k))
    return "GOP %d (seed %d, pan %.1f, noise %.1f, qscale %d, f_code %d): %d pictures, %d bytes (first %d), %.0f s" % (
        k, seed, pan, noise, q, f, n, len(es), int(offs[1]) if len(offs) > 1 else len(es), time.time() - t0)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(len(GOPS)))      # e.g. 8,9,10,11
    with mp.Pool(min(len(which), os.cpu_count() or 1)) as pool:
        for line in pool.imap_unordered(one, which):
            print(line, flush=True)
