#!/usr/bin/env python3
"""Pictures AND sound of N live streams on one GPU: a live video handle (C ABI part 5) and a live audio handle (part 6) fed every
tick with a picture and a frame per stream -- the audio tick run (a) after the video tick, (b) BETWEEN the video tick's two halves
(jsmpeg_hip_live_tick_begin, jsmpeg_hip_mp2_live_tick, jsmpeg_hip_live_tick_end: the audio pass rides beside the slice parse on CUs
the parse leaves free).  Host clock per tick, writes included; every picture's device hash and every frame's samples against the oracle.

    python tools/live_av_bench.py [--streams 64] [--pictures 37]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def run(streams=64, pictures=37, config="cfg2_1080p"):
    from jsmpeg_amd import build, cabi, hashing, live as jl, mp2, synth
    import live_bench
    oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    W, H = {"cfg2_1080p": (1920, 1080), "cfg1_720p": (1280, 720)}[config]
    gen = [synth.generate_config(config, n_frames=pictures, stream=s) for s in range(streams)]
    vwrites = [live_bench.picture_writes(es, offs) for es, offs in gen]
    amade = [synth.generate_mp2_config("mp2_stereo_44k_192", pictures, stream=s) for s in range(streams)]
    ab = [[int(o) for o in offs] + [len(d)] for d, offs in amade]
    awrites = [[amade[s][0][ab[s][k]:ab[s][k + 1]] for k in range(pictures)] for s in range(streams)]
    from concurrent.futures import ThreadPoolExecutor

    def one(s):
        frames, _, _ = cabi.decode_stream(oracle, gen[s][0], keep="planes")
        return [hashing.frame_hash(*f) for f in frames]
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        vwant = list(ex.map(one, range(streams)))
    awant = [cabi.decode_mp2_stream(oracle, d)[0].view(np.uint32) for d, _ in amade]
    store = max(512 * 1024, 2 * max(len(w) for ws in vwrites for w in ws))
    out = {"streams": streams, "ticks": pictures, "config": config}
    for name in ("audio_after_the_video_tick", "audio_between_the_video_ticks_halves"):
        ms, bad = [], 0
        with jl.Live(W, H, streams, pictures_per_tick=1, store_bytes=store) as lv, mp2.Mp2Live(streams, max_frames_per_tick=2) as al:
            for _ in range(streams):
                assert lv.open() == al.open()
            for k in range(pictures):
                t0 = time.perf_counter()
                for s in range(streams):
                    lv.write(s, vwrites[s][k], pts=k / 30.0)
                    al.write(s, k / 30.0, awrites[s][k])
                if name == "audio_after_the_video_tick":
                    nv = lv.tick(flush=True)
                    na = al.tick()
                else:
                    lv.tick_begin(flush=True)
                    na = al.tick()
                    nv = lv.tick_end()
                ms.append((time.perf_counter() - t0) * 1e3)
                hs = lv.frame_hashes()
                pcm = al.read_pcm().view(np.uint32)
                if nv != streams or na != streams:
                    raise RuntimeError("tick %d: %d pictures, %d frames of %d streams" % (k, nv, na, streams))
                for i, p in enumerate(lv.pictures()):
                    bad += int(hs[i]) != vwant[p.stream][k]
                for i, f in enumerate(al.frames()):
                    bad += not np.array_equal(pcm[i], awant[f["stream"]][k])
        p_ticks = [m for k, m in enumerate(ms) if k % 12 and k > 2]
        out[name] = {"ms_per_tick_p_pictures": round(float(np.median(p_ticks)), 4), "differing_from_oracle": bad}
        if bad:
            raise RuntimeError("PARITY FAILURE: %d pictures / frames differ from the oracle (%s)" % (bad, name))
    out["note"] = ("a picture and an MP2 frame per stream per tick, 2 x %d writes + the two ticks on the host clock (median of the ticks of P pictures); every picture's "
                   "hash and every frame's samples == the oracle's" % streams)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--pictures", type=int, default=37)
    a = ap.parse_args()
    print(json.dumps(run(a.streams, a.pictures)))
