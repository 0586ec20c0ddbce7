#!/usr/bin/env python3
"""Turn the rocprofv3 (rocpd sqlite) outputs under gpurun_out/prof/ into the small text summaries committed under
profiles/: per-kernel time statistics of the --kernel-trace --stats run and per-kernel PMC counter sums/averages."""
import glob
import os
import sqlite3
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "round1"
cmd_trace = sys.argv[3] if len(sys.argv) > 3 else "python bench.py --steps 1 --warmup 1 --cpu-sample 0"
cmd_pmc = sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 1 --warmup 0 --cpu-sample 0 --reads 16000"
os.makedirs("profiles", exist_ok=True)

db = sqlite3.connect(os.path.join(src, "trace", "bench_results.db"))
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
                  "max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
with open("profiles/%s_kernel_stats.csv" % tag, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- %s   (durations in ns)\n" % cmd_trace)
    f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent,vgpr,sgpr,lds_bytes,max_grid_x,workgroup_x\n")
    for r in rows:
        f.write("\"%s\",%d,%d,%.0f,%d,%d,%.3f,%d,%d,%d,%d,%d\n" % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10]))
print(open("profiles/%s_kernel_stats.csv" % tag).read())

with open("profiles/%s_pmc.csv" % tag, "w") as f:
    f.write("# rocprofv3 --pmc <counters> -- %s  (one pass per counter group)\n" % cmd_pmc)
    f.write("pass,kernel,counter,dispatches,sum,avg_per_dispatch,avg_dispatch_ns\n")
    for d in sorted(glob.glob(os.path.join(src, "pmc_*", "bench_results.db"))):
        name = os.path.basename(os.path.dirname(d))
        c = sqlite3.connect(d)
        for r in c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
                           "where kernel_name like 'void k_%' or kernel_name like 'k_%' group by kernel_name, counter_name order by kernel_name, counter_name"):
            f.write("%s,\"%s\",%s,%d,%.6g,%.6g,%.0f\n" % (name, r[0], r[1], r[2], r[3], r[4], r[5]))
print(open("profiles/%s_pmc.csv" % tag).read())
