#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.x) rocpd sqlite database as text: per-kernel calls / avg / min / max
duration, launch geometry and register usage; PMC counters (summed per kernel) if present.
usage: rocprof_summary.py <results.db> [> profiles/xxx.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("# kernel summary (durations in microseconds)")
print("calls\tavg_us\tmin_us\tmax_us\ttotal_ms\tgrid\twg\tlds\tscratch\tvgpr\tagpr\tsgpr\tname")
q = ("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, sum(end-start)/1e6, "
     "max(grid_x), max(workgroup_x), max(lds_size), max(scratch_size), max(vgpr_count), max(accum_vgpr_count), "
     "max(sgpr_count) from kernels group by name order by sum(end-start) desc")
for r in cur.execute(q):
    name = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
    print("%d\t%.1f\t%.1f\t%.1f\t%.3f\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%s" % (r[1], r[2], r[3], r[4], r[5], *r[6:13], name))
try:
    rows = list(cur.execute("select k.name, p.name, count(*), sum(e.value), avg(e.value) from pmc_events e "
                            "join pmc_info p on e.pmc_id = p.id join kernels k on e.event_id = k.id "
                            "group by k.name, p.name order by k.name"))
except sqlite3.Error:
    rows = []
    try:
        cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
        if cols:
            print("\n# counters_collection columns:", cols)
            namecol = "kernel_name" if "kernel_name" in cols else "name"
            rows = list(cur.execute(f"select {namecol}, counter_name, count(*), sum(value), avg(value) from "
                                    f"counters_collection group by {namecol}, counter_name"))
    except sqlite3.Error as e:
        print("# no PMC data:", e)
if rows:
    print("\n# PMC counters: kernel, counter, dispatches, sum, avg per dispatch")
    for r in rows:
        print("%s\t%s\t%d\t%.6g\t%.6g" % ((r[0][:70],) + tuple(r[1:])))
