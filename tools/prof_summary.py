"""Summarise a rocprofv3 results .db (kernel trace): per-kernel count / avg / total, like --stats."""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
d = collections.defaultdict(list)
for n, s, e in rows:
    d[n].append((e - s) / 1000.0)
tot = sum(sum(v) for v in d.values())
print("%-110s %8s %10s %9s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "med_us", "pct"))
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-110s %8d %10.1f %9.3f %9.3f %6.2f" % (n[:110], len(v), sum(v), sum(v) / len(v), v2[len(v) // 2], 100 * sum(v) / tot))
