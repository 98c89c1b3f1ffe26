"""Per-kernel durations of a rocprofv3 --kernel-trace run of tools/live_bench.py, split by launch size (the live tick's kernels and
the one-picture ABI's run in the same process: a tick of 64 x 1080p P pictures parses 4352 slices in 1088 wavefronts, grid
69632; a decode() of one picture 69 slices in 69, grid 35328) -> profiles/<tag>_live_kernel_stats.txt.
    cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d <repo>/gpurun_out/prof_live -- python <repo>/tools/live_bench.py --streams 64 --pictures 61 --no-check
    python tools/live_kernel_stats.py r06 gpurun_out/prof_live
(own `timeout`: on this pool rocprofv3 did not come back after the program had finished; the .db is complete by then)"""
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d = sys.argv[1:3]
c = sqlite3.connect(sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))[-1])
rows = list(c.execute("select name, grid_x, count(*), avg(duration), min(duration), max(duration) from kernels where name like 'k_%' "
                      "group by name, grid_x order by name, grid_x"))
out = os.path.join(ROOT, "profiles", "%s_live_kernel_stats.txt" % tag)
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace -- python tools/live_bench.py --streams 64 --pictures 61 --no-check   (61 ticks of 64 x 1080p: 55 of P pictures,\n")
    f.write("# 6 of I pictures; then 244 decode() calls of the one-picture ABI, 4 decoders in turn).  Microseconds; by kernel and launch size.\n")
    f.write("%-16s %10s %6s %10s %10s %10s\n" % ("kernel", "grid_x", "calls", "avg_us", "min_us", "max_us"))
    for name, grid, n, avg, lo, hi in rows:
        f.write("%-16s %10d %6d %10.1f %10.1f %10.1f\n" % (name.split("(")[0][:16], grid, n, avg / 1e3, lo / 1e3, hi / 1e3))
print(open(out).read())
