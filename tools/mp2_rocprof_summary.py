"""Turns the rocprofv3 (rocpd sqlite) outputs of tools/final_profile.sh into the tracked summary
profiles/<tag>_mp2_kernel_stats.txt: per-kernel time of the MP2 stage (`--kernel-trace --stats`), register / LDS use,
and HBM bytes per launch from separate FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE doubled: gfx950 reports half of
streamed reads, MI355X_MICROARCH.md section HBM -- same correction as tools/rocprof_summary.py).
    python tools/mp2_rocprof_summary.py r01i gpurun_out/prof_mp2 gpurun_out/prof_mp2_fetch gpurun_out/prof_mp2_write
"""
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d_trace, d_fetch, d_write = sys.argv[1:5]


def db(d):
    return sqlite3.connect(sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))[-1])


def short(n):
    return n.split("(")[0]


out = ["# rocprofv3 --kernel-trace --stats -- python tools/mp2_bench.py --reps 20   (MP2 audio stage: 64 streams x 154 frames per pass)",
       "# (durations in microseconds; 21 decode passes = 1 warm-up + 20 timed)",
       "%-28s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
for name, calls, total, avg, pct in db(d_trace).execute("select * from top_kernels"):
    out.append("%-28s %8d %14.1f %12.2f %7.2f%%" % (short(name)[:28], calls, total, avg, pct))
out += ["", "# per-dispatch register / LDS / scratch usage"]
for row in db(d_trace).execute("select name, max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                               "max(workgroup_x), min(grid_x), max(grid_x) from kernels group by name"):
    out.append("%-28s vgpr %3d sgpr %3d lds %6d scratch %d wg %d grid %d..%d" % ((short(row[0])[:28],) + row[1:]))
try:
    q = ("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name")
    fetch = {short(r[0]): r[1:] for r in db(d_fetch).execute(q, ("FETCH_SIZE",))}
    write = {short(r[0]): r[1:] for r in db(d_write).execute(q, ("WRITE_SIZE",))}
    out += ["", "# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs), KiB per dispatch;",
            "# hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950 FETCH_SIZE counts half)",
            "%-16s %6s %14s %14s %16s" % ("kernel", "n", "fetch_avg_KiB", "write_avg_KiB", "hbm_MB_per_launch")]
    for k in sorted(set(fetch) | set(write)):
        fa, wa = fetch.get(k, (0, 0)), write.get(k, (0, 0))
        out.append("%-16s %6d %14.1f %14.1f %16.2f" % (k[:16], fa[0], fa[1], wa[1], (2 * fa[1] + wa[1]) * 1024 / 1e6))
except Exception as e:
    out.append("# counter passes not summarised: %r" % (e,))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", "%s_mp2_kernel_stats.txt" % tag), "w").write("\n".join(out) + "\n")
print("\n".join(out))
