#!/usr/bin/env python3
"""overlap.py <rocprofv3 results.db> -- do kernels of different streams actually run concurrently?
Prints, per kernel name, total time and the share of it during which some OTHER kernel was also running."""
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if t.startswith("kernels")]
rows = list(cur.execute(f"select name, start, end, stream_id from {kt[0]} order by start")) if "stream_id" in [c[1] for c in cur.execute(f"pragma table_info({kt[0]})")] \
    else [(n, s, e, 0) for n, s, e in cur.execute(f"select name, start, end from {kt[0]} order by start")]
rows = [(n.split("(")[0].replace("void ", ""), s, e, q) for n, s, e, q in rows if "k_ks" in n]
tot = collections.Counter(); ov = collections.Counter()
for i, (n, s, e, q) in enumerate(rows):
    tot[n] += e - s
    # union of overlaps with other kernels
    segs = sorted((max(s, s2), min(e, e2)) for j, (n2, s2, e2, q2) in enumerate(rows) if j != i and s2 < e and e2 > s)
    cov, last = 0, s
    for a, b in segs:
        a = max(a, last)
        if b > a: cov += b - a; last = b
    ov[n] += cov
span = rows[-1][2] - rows[0][1]
print(f"{len(rows)} keyswitch kernels over {span/1e6:.3f} ms; sum of kernel times {sum(tot.values())/1e6:.3f} ms")
for n in tot:
    print(f"  {n:40s} {tot[n]/1e6:8.3f} ms   concurrent with another kernel {100*ov[n]/tot[n]:5.1f} %")
