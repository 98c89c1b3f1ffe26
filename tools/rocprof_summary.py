"""Turns the rocprofv3 (rocpd sqlite) outputs that a gpurun call left under
gpurun_out/ into the small tracked summaries under profiles/:
  profiles/<tag>_kernel_stats.txt   -- `--kernel-trace --stats` per-kernel table
  profiles/<tag>_pmc.txt            -- FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
  profiles/pmc_traffic.json         -- HBM bytes per launch, per kernel (read by bench.py)
FETCH_SIZE is doubled (gfx950 reports half of streamed reads, MI355X_MICROARCH.md
section HBM; checked here on k_hash, which reads a known byte count).
    python tools/rocprof_summary.py r01 gpurun_out/prof_trace gpurun_out/prof_fetch gpurun_out/prof_write
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d_trace, d_fetch, d_write = sys.argv[1:5]
d_valu = sys.argv[5] if len(sys.argv) > 5 else None      # optional fourth pass: --pmc SQ_INSTS_VALU SQ_WAVES


def db(d):
    return sqlite3.connect(sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))[-1])


def short(n):
    return n.split("(")[0]


os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "%s_kernel_stats.txt" % tag), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --no-napi --no-h2d\n")
    f.write("# (durations in microseconds; 6 decode passes = 1 warm-up + 5 timed; a step that reconstructs level by level -- wide batches, dense\n")
    f.write("#  intra pictures -- has per pass one k_recon launch per predicted level + one k_recon_intra[_dense] launch for the level without\n")
    f.write("#  a forward reference: the step's reconstruct = their sum; bench.py's roofline figures are per launch = that / launches)\n")
    f.write("%-28s %8s %14s %12s %8s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in db(d_trace).execute("select * from top_kernels"):
        f.write("%-28s %8d %14.1f %12.2f %7.2f%%\n" % (short(name)[:28], calls, total, avg, pct))
    f.write("\n# per-dispatch register / LDS / scratch usage\n")
    for row in db(d_trace).execute("select name, max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                                   "max(workgroup_x), min(grid_x), max(grid_x) from kernels group by name"):
        f.write("%-28s vgpr %3d sgpr %3d lds %6d scratch %d wg %d grid %d..%d\n" % ((short(row[0])[:28],) + row[1:]))

q = ("select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
     "where counter_name = ? group by kernel_name")
fetch = {short(r[0]): r[1:] for r in db(d_fetch).execute(q, ("FETCH_SIZE",))}
write = {short(r[0]): r[1:] for r in db(d_write).execute(q, ("WRITE_SIZE",))}
traffic = {}
with open(os.path.join(ROOT, "profiles", "%s_pmc.txt" % tag), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs), KiB per dispatch\n")
    f.write("# hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950 FETCH_SIZE counts half)\n")
    f.write("%-16s %6s %14s %14s %14s %16s\n" % ("kernel", "n", "fetch_avg_KiB", "fetch_max_KiB", "write_avg_KiB", "hbm_MB_per_launch"))
    for k in sorted(set(fetch) | set(write)):
        fa = fetch.get(k, (0, 0, 0, 0))
        wa = write.get(k, (0, 0, 0, 0))
        hbm = (2 * fa[1] + wa[1]) * 1024
        traffic[k] = int(hbm)
        f.write("%-16s %6d %14.1f %14.1f %14.1f %16.1f\n" % (k[:16], fa[0], fa[1], fa[3], wa[1], hbm / 1e6))
tj = {k: v for k, v in traffic.items() if k.startswith("k_")}
tj["source"] = "profiles/%s_pmc.txt" % tag
tj["fetch_bytes"] = {k: int(2 * fetch[k][1] * 1024) for k in fetch if k.startswith("k_")}
tj["write_bytes"] = {k: int(write[k][1] * 1024) for k in write if k.startswith("k_")}
json.dump(tj, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
if d_valu and glob.glob(os.path.join(d_valu, "**", "*.db"), recursive=True):
    valu = {short(r[0]): r[1:] for r in db(d_valu).execute(q, ("SQ_INSTS_VALU",))}
    waves = {short(r[0]): r[1:] for r in db(d_valu).execute(q, ("SQ_WAVES",))}
    with open(os.path.join(ROOT, "profiles", "%s_pmc.txt" % tag), "a") as f:
        f.write("\n# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES (its own run): wavefront-instructions per dispatch\n")
        f.write("%-16s %6s %16s %12s %12s\n" % ("kernel", "n", "valu_instr_avg", "waves_avg", "valu_per_wave"))
        for k in sorted(valu):
            w = waves.get(k, (0, 0, 0, 0))[1]
            f.write("%-16s %6d %16.0f %12.0f %12.1f\n" % (k[:16], valu[k][0], valu[k][1], w, valu[k][1] / w if w else 0))
    vj = {k: int(v[1]) for k, v in valu.items() if k.startswith("k_")}
    vj["source"] = "profiles/%s_pmc.txt" % tag
    json.dump(vj, open(os.path.join(ROOT, "profiles", "pmc_valu.json"), "w"), indent=1)
print(open(os.path.join(ROOT, "profiles", "%s_kernel_stats.txt" % tag)).read())
print(open(os.path.join(ROOT, "profiles", "%s_pmc.txt" % tag)).read())
