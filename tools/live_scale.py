"""How many LIVE 1080p streams one GPU carries: tools/live_bench.py's tick (a picture per stream per tick, every picture checked
against the oracle fed the same writes) at growing stream counts.  A 30 frames/s stream needs a tick every 33.3 ms; the table
says what a tick of S streams takes and what share of that budget it is.

    python tools/live_scale.py [--streams 64,128,256,512,1024] [--pictures 13] [--json profiles/rNN_live_scale.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import live_bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="64,128,256,512,1024")
    ap.add_argument("--pictures", type=int, default=13)
    ap.add_argument("--config", default="cfg2_1080p")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--overlap", action="store_true", help="the next tick's pictures are written while a tick is on the device (tick_begin / tick_end)")
    ap.add_argument("--json")
    a = ap.parse_args()
    rows, bad = [], 0
    for s in [int(x) for x in a.streams.split(",")]:
        r = live_bench.run(s, a.pictures, a.config, 1, check=not a.no_check, abi_streams=0, verbose=False, overlap=a.overlap)
        bad += r.get("pictures_differing_from_oracle") or 0
        row = dict(streams=s, ms_per_tick_p=r["ms_per_tick_p_pictures"], ms_per_tick_i=r["ms_per_tick_i_pictures"], ms_writes=r["ms_writes_median"],
                   pictures_per_s=r["pictures_per_s"], share_of_a_30_fps_tick=(r["ms_per_tick_p_pictures"] or 0) / (1000.0 / 30),
                   differing=r.get("pictures_differing_from_oracle"), parts_p=r["parts_ms_p_tick"], parts_i=r["parts_ms_i_tick"])
        rows.append(row)
        print("%5d streams: %.3f ms per tick of P pictures (writes %.3f, parse %.3f, reconstruct %.3f), %.3f per tick of I pictures; %.0f pictures/s; "
              "%.1f %% of a 30 frames/s tick; differing from the oracle: %s"
              % (s, row["ms_per_tick_p"], row["ms_writes"], row["parts_p"]["parse_ms"], row["parts_p"]["recon_ms"], row["ms_per_tick_i"] or -1,
                 row["pictures_per_s"], 100 * row["share_of_a_30_fps_tick"], row["differing"]), flush=True)
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump(dict(config=a.config, pictures_per_stream=a.pictures, writes_beside_the_tick=a.overlap, rows=rows), open(a.json, "w"), indent=1)
    sys.exit(1 if bad else 0)
