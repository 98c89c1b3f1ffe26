"""Kernel-timing experiments (no parity gate, no torch): decode the cfg2 batch a few times and print the engine's
phase timings.  For -D experiment builds whose output is deliberately wrong.
    python tools/kbench.py [streams] [frames] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = sys.argv[:1] + sys.argv[1:]
import bench  # noqa: E402  (generate_streams only)
bench.CONFIG = os.environ.get("JSMPEG_KBENCH_CONFIG", bench.CONFIG)   # e.g. cfg1_720p, cfg4_2160p: the other configurations as timings
from jsmpeg_amd import batch as jb, synth  # noqa: E402

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 120
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
warm = 2 if reps > 2 else 1      # the HIP runtime spends ~8 ms once inside the SECOND pass of a process (seen in the host trace): not a figure of the path
cfg = synth.CONFIGS[bench.CONFIG]
for kv in filter(None, os.environ.get("JSMPEG_SYNTH_OVERRIDES", "").split(",")):     # e.g. gop=6,ac_max=8
    cfg[kv.split("=")[0]] = int(kv.split("=")[1])
if os.environ.get("JSMPEG_SYNTH_MV_JITTER"):       # coherent motion: one vector per picture, +-(k - 1) of jitter per macroblock
    cfg["mv_jitter"] = int(os.environ["JSMPEG_SYNTH_MV_JITTER"])
gen = bench.generate_streams(0, n_streams, frames)
streams = [g[0] for g in gen]
total = sum(len(s) for s in streams)
def phase_clk():
    """a JM_T_PHASECLK timing build (tools/variants.sh): the clocks wavefront 0 of every k_recon workgroup spent per phase since the last call"""
    import ctypes
    from jsmpeg_amd import build
    lib = ctypes.CDLL(build.LIB_HIP)
    if not hasattr(lib, "jsmpeg_hip_debug_phase_clk"):
        return None
    out = (ctypes.c_ulonglong * 8)()
    return list(out) if lib.jsmpeg_hip_debug_phase_clk(out) == 0 else None


def run(order):
    if order is not None:
        os.environ["JSMPEG_HIP_RECON_ORDER"] = order
    with jb.Batch(cfg["width"], cfg["height"], n_streams, n_streams * frames + 8, total + 64 * n_streams + 4096) as b:
        b.upload(streams)
        acc = None
        for r in range(reps):
            if r == reps - 1:
                phase_clk()      # (reset)
            b.decode()
            t = b.timings()
            if r >= warm:
                acc = t if acc is None else {k: acc[k] + t[k] for k in t}
        lv = b.counters()["levels"]
        pc = phase_clk()
        if pc:
            names = ("first look", "(barrier)", "scatter + predict", "(barrier)", "transform", "(barrier)", "pixels + stores issued", "stores acknowledged")
            print("wavefront 0's clocks of the last pass by phase (JM_T_PHASECLK), share of the sum:",
                  ", ".join("%s %.1f %%" % (n, 100.0 * v / max(1, sum(pc))) for n, v in zip(names, pc)), "| sum %.3e" % sum(pc))
        try:
            lt = b.level_timings()
            print("recon launches ms:", " ".join("%.3f" % x for x in lt))
        except Exception as e:  # an older library under JSMPEG_HIP_LIB
            print("no level timings:", e)
        try:
            print("reconstruct:", b.recon_info())
        except Exception as e:
            print("no recon info:", e)
        ms = acc["total_ms"] / (reps - warm)
        print(("order %s: " % order if order is not None else "") + str({k: round(v / (reps - warm), 3) for k, v in acc.items()}),
              "recon per level %.3f" % (acc["recon_ms"] / (reps - warm) / lv),
              "| %s %d x %d: %.0f frames/s, %.0f Mpixel/s" % (bench.CONFIG, n_streams, frames, n_streams * frames / ms * 1e3,
                                                             n_streams * frames / ms * 1e3 * cfg["width"] * cfg["height"] / 1e6), flush=True)


# JSMPEG_KBENCH_ORDERS=0,1,2,4: one batch per value of JSMPEG_HIP_RECON_ORDER (streams a class walks in lockstep; 0 = level by level)
orders = os.environ.get("JSMPEG_KBENCH_ORDERS")
for o in (orders.split(",") if orders else [None]):
    run(o)
