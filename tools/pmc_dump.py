"""Prints the per-kernel averages of every counter found in rocprofv3 --pmc outputs (rocpd sqlite) under the given
directories.  python tools/pmc_dump.py gpurun_out/pmc_a gpurun_out/pmc_b ..."""
import glob
import os
import sqlite3
import sys

for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(path)
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
             "group by kernel_name, counter_name order by kernel_name, counter_name")
        print("#", path)
        for k, c, n, v in db.execute(q):
            k = k.split("(")[0]
            if k.startswith("k_"):
                print("%-14s %-26s n=%-4d avg=%.4g" % (k, c, n, v))
