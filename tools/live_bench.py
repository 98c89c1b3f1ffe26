"""LIVE streams (include/jsmpeg_hip.h part 5) as a timing: S concurrent 1920x1080 streams, every tick brings each stream
one picture (a write() per stream, as ts.js delivers them) and ONE jsmpeg_hip_live_tick decodes them all -- ms per tick and
pictures per second on the host clock (writes included), beside the one-picture ABI driven the same way (a decoder per
stream: write a picture, decode()).  Every picture's device hash is checked against the oracle's decoder fed the same writes.

    python tools/live_bench.py [--streams 64] [--pictures 36] [--config cfg2_1080p] [--pictures-per-tick 1] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jsmpeg_amd import build, cabi, hashing, live as jl, synth  # noqa: E402


def picture_writes(es, offs):
    n = len(offs) - 1
    return [es[int(offs[k]):(len(es) if k == n - 1 else int(offs[k + 1]))] for k in range(n)]


def run(streams=64, pictures=36, config="cfg2_1080p", per_tick=1, check=True, abi_streams=4, width=None, height=None, verbose=True, via_node=False, overlap=False, also_overlapped=False, _want=None):
    say = print if verbose else (lambda *a, **k: None)
    kw = {}
    if width:
        kw = dict(width=width, height=height)
    gen = [synth.generate_config(config, n_frames=pictures, stream=s, **kw) for s in range(streams)]
    W, H = (width, height) if width else {"cfg2_1080p": (1920, 1080), "cfg1_720p": (1280, 720), "cfg4_2160p": (3840, 2160), "cfg0_240p_intra": (320, 240)}[config]
    writes = [picture_writes(es, offs) for es, offs in gen]
    biggest = max(len(w) for ws in writes for w in ws)
    store = max(512 * 1024, 2 * per_tick * biggest)
    want = _want
    if check and want is None:
        oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
        t0 = time.perf_counter()
        from concurrent.futures import ThreadPoolExecutor

        def one(s):
            frames, _, _ = cabi.decode_stream(oracle, gen[s][0], keep="planes")
            return [hashing.frame_hash(*f) for f in frames]
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            want = list(ex.map(one, range(streams)))
        say("oracle: %d streams x %d pictures in %.1f s" % (streams, pictures, time.perf_counter() - t0))
    out = dict(writes_beside_the_tick=bool(overlap), config=config, streams=streams, pictures_per_stream=pictures, pictures_per_stream_per_tick=per_tick, width=W, height=H,
               bytes_per_picture=int(np.mean([len(w) for ws in writes for w in ws])))
    with jl.Live(W, H, streams, pictures_per_tick=max(per_tick, 1), store_bytes=store) as lv:
        ids = [lv.open() for _ in range(streams)]
        ticks, bad, n_out = [], 0, 0
        parts = []
        got = [[] for _ in range(streams)]
        def feed(k):
            for s in range(streams):
                for w in writes[s][k:k + per_tick]:
                    lv.write(ids[s], w, pts=k / 30.0)
        if overlap:
            feed(0)                                   # (overlapped: tick k's pictures are written while tick k - 1 is on the device)
        for k in range(0, pictures, per_tick):
            t0 = time.perf_counter()
            if overlap:
                lv.tick_begin(flush=True)
                tb = time.perf_counter()
                feed(k + per_tick)
                t1 = time.perf_counter()
                n = lv.tick_end()
                t2 = time.perf_counter()
                ticks.append((t2 - t0, t1 - tb, (tb - t0) + (t2 - t1), n))
            else:
                feed(k)
                t1 = time.perf_counter()
                n = lv.tick(flush=True)
                t2 = time.perf_counter()
                ticks.append((t2 - t0, t1 - t0, t2 - t1, n))
            parts.append(lv.timings())
            n_out += n
            if check:
                hs = lv.frame_hashes()
                for i, p in enumerate(lv.pictures()):
                    got[p.stream].append(int(hs[i]))
        if check:
            for s in range(streams):
                bad += sum(1 for a, b in zip(got[ids[s]], want[s]) if a != b) + abs(len(got[ids[s]]) - len(want[s]))
        # the first ticks carry the I pictures (every stream starts a GOP at once): report the steady state apart
        all_ms = np.array([t[0] for t in ticks]) * 1e3
        gop = 12 // per_tick if per_tick <= 12 else 1
        p_ticks = [i for i in range(len(ticks)) if gop and i % gop != 0 and i > 0] or list(range(len(ticks)))
        i_ticks = [i for i in range(len(ticks)) if gop and i % gop == 0 and i > 0]
        med = lambda xs: float(np.median(xs)) if len(xs) else None
        out.update(ticks=len(ticks), pictures=n_out, pictures_differing_from_oracle=bad if check else None,
                   ms_per_tick_median=med(all_ms[1:]), ms_per_tick_max=float(all_ms[1:].max()) if len(all_ms) > 1 else None,
                   ms_per_tick_each=[round(float(x), 3) for x in all_ms], ms_writes_each=[round(t[1] * 1e3, 3) for t in ticks], ms_per_tick_p_pictures=med(all_ms[p_ticks]), ms_per_tick_i_pictures=med(all_ms[i_ticks]),
                   ms_writes_median=med([ticks[i][1] * 1e3 for i in p_ticks]), ms_tick_call_median=med([ticks[i][2] * 1e3 for i in p_ticks]),
                   pictures_per_s=n_out / sum(t[0] for t in ticks[1:]) * (len(ticks) - 1) / len(ticks) if len(ticks) > 1 else None,
                   parts_ms_p_tick={k2: med([parts[i][k2] for i in p_ticks]) for k2 in parts[0]},
                   parts_ms_i_tick={k2: med([parts[i][k2] for i in i_ticks]) for k2 in parts[0]} if i_ticks else None)
        # pictures per second over everything but the first tick (allocation, first-touch)
        tot = sum(t[0] for t in ticks[1:])
        out["pictures_per_s"] = sum(t[3] for t in ticks[1:]) / tot if tot > 0 else None
    say("live%s: %d streams, %d picture(s) per stream per tick: %.3f ms per tick of P pictures (writes %.3f + tick %.3f), %.3f per tick of I pictures; "
        "%.0f pictures/s overall; differing from the oracle: %s" % (" (writes beside the tick in flight: tick = its two calls)" if overlap else "", streams, per_tick, out["ms_per_tick_p_pictures"] or -1, out["ms_writes_median"] or -1, out["ms_tick_call_median"] or -1,
                                                                   out["ms_per_tick_i_pictures"] or -1, out["pictures_per_s"] or -1, out["pictures_differing_from_oracle"]))
    say("      parts of a P tick (ms):", {k2: round(v, 3) for k2, v in out["parts_ms_p_tick"].items()})
    say("      every tick (ms):", out["ms_per_tick_each"])
    say("      its writes (ms):", out["ms_writes_each"])
    worst = int(np.argmax(all_ms[1:])) + 1 if len(all_ms) > 1 else 0
    say("      the slowest tick (%d) in parts:" % worst, {k2: round(v, 3) for k2, v in parts[worst].items()})
    # the one-picture ABI driven the same way: a decoder per stream, a picture written, a picture decoded (planes to the host)
    if abi_streams:
        k_abi = min(abi_streams, streams)
        decs = [cabi.Mpeg1Decoder(build.LIB_HIP, 1 << 20, cabi.MODE_EVICT) for _ in range(k_abi)]
        ts = []
        for k in range(pictures):
            for s in range(k_abi):
                t0 = time.perf_counter()
                decs[s].write(writes[s][k])
                while decs[s].decode():
                    pass
                ts.append((time.perf_counter() - t0, k % 12 == 0))
        for d in decs:
            d.close()
        p_ms = float(np.median([t for t, i in ts[k_abi:] if not i])) * 1e3
        tot = sum(t for t, _ in ts[k_abi:])
        out["one_picture_abi"] = dict(streams=k_abi, ms_per_p_picture=p_ms, pictures_per_s=(len(ts) - k_abi) / tot)
        out["live_over_one_picture_abi"] = out["pictures_per_s"] / out["one_picture_abi"]["pictures_per_s"] if out["pictures_per_s"] else None
        say("one-picture ABI (write a picture, decode(), planes to the host), %d decoders in turn: %.3f ms per P picture, %.0f pictures/s -> live tick = %.1f x"
            % (k_abi, p_ms, out["one_picture_abi"]["pictures_per_s"], out["live_over_one_picture_abi"] or -1))
    if also_overlapped and not overlap:
        # the same ticks with tick k + 1's pictures written while tick k is on the device (jsmpeg_hip_live_tick_begin / _end)
        r2 = run(streams, pictures, config, per_tick, check=check, abi_streams=0, width=width, height=height, verbose=verbose, overlap=True, _want=want)
        out["writes_beside_the_tick_in_flight"] = {k2: r2[k2] for k2 in ("ms_per_tick_p_pictures", "ms_per_tick_i_pictures", "ms_per_tick_median", "ms_per_tick_max", "ms_writes_median",
                                                                         "ms_tick_call_median", "pictures_per_s", "pictures_differing_from_oracle", "pictures")}
        if out.get("one_picture_abi") and r2["pictures_per_s"]:
            out["writes_beside_the_tick_in_flight"]["over_one_picture_abi"] = r2["pictures_per_s"] / out["one_picture_abi"]["pictures_per_s"]
    if via_node and want is not None:
        out["via_napi"] = node_run(gen, want, streams, W, H)
        if "ms_per_tick_p_pictures" in out["via_napi"]:
            say("  the same from Node (JSMpeg.HIPLive over the N-API addon): %.3f ms per tick of P pictures, %.0f pictures/s"
                % (out["via_napi"]["ms_per_tick_p_pictures"], out["via_napi"]["pictures_per_s"]))
            for key, what in (("writes_beside_the_tick_in_flight", "writes beside the tick (tickBegin / tickEnd)"), ("with_planes_to_host", "every picture's planes to the host as well (onFrame)"),
                              ("with_planes_to_host_pipelined", "... the planes of tick k travelling beside the pass of tick k + 1 ({pipelined: true})")):
                r = out["via_napi"].get(key)
                if r:
                    say("      %s: %.3f ms per tick of P pictures, %.0f pictures/s" % (what, r["ms_per_tick_p_pictures"], r["pictures_per_s"]))
                    if r.get("wall"):
                        say("          by the loop's wall clock, nothing between the ticks: %.3f ms per tick, %.0f pictures/s" % (r["wall"]["ms_per_tick"], r["wall"]["pictures_per_s"]))
    return out


def node_run(gen, want, streams, W, H):
    """the same ticks from the host north_star names: tools/live_bench_node.js (JSMpeg.HIPLive), hashes against the same oracle vectors"""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("node") or not os.path.exists(os.path.join(ROOT, "jsmpeg_amd", "js", "jsmpeg_hip.node")):
        return {"error": "node or the addon is not there"}
    td = tempfile.mkdtemp(prefix="jsmpeg_live_")
    try:
        for s in range(streams):
            gen[s][0].tofile(os.path.join(td, "s%d.m1v" % s))
        json.dump([[int(x) for x in gen[s][1]] for s in range(streams)], open(os.path.join(td, "offsets.json"), "w"))
        json.dump({str(s): ["%016x" % h for h in want[s]] for s in range(streams)}, open(os.path.join(td, "hashes.json"), "w"))
        p = subprocess.run(["node", os.path.join(ROOT, "tools", "live_bench_node.js"), "--dir", td, "--streams", str(streams), "--width", str(W), "--height", str(H)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else {"error": "no result (rc %d): %s" % (p.returncode, p.stderr.decode()[-300:])}
    finally:
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--pictures", type=int, default=36)
    ap.add_argument("--config", default="cfg2_1080p")
    ap.add_argument("--pictures-per-tick", type=int, default=1)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--node", action="store_true", help="also run the same ticks from Node (tools/live_bench_node.js)")
    ap.add_argument("--overlap", action="store_true", help="tick k + 1's pictures are written between jsmpeg_hip_live_tick_begin and _end of tick k")
    ap.add_argument("--both", action="store_true", help="the plain ticks, then the same with the writes beside the tick in flight")
    ap.add_argument("--json")
    a = ap.parse_args()
    res = run(a.streams, a.pictures, a.config, a.pictures_per_tick, check=not a.no_check, via_node=a.node, overlap=a.overlap, also_overlapped=a.both)
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump(res, open(a.json, "w"), indent=1)
    sys.exit(1 if res.get("pictures_differing_from_oracle") or (res.get("writes_beside_the_tick_in_flight") or {}).get("pictures_differing_from_oracle") else 0)
